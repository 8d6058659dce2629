"""The randomised sweeps of tests/sweeps (GPU result against the oracle, unit by unit, over
configurations drawn from a seed) with one fixed seed and a few configurations each, so that
they run with the suite; tests/sweeps/run_all.sh SEED runs them at length."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

# script, arguments after the seed (number of configurations, time budget where the script takes one)
SWEEPS = [
    ('sweep_fit.py', ['10', '40']),
    ('sweep_binning.py', ['12']),
    ('sweep_exposure.py', ['16']),
    ('sweep_spectra_blocks.py', []),
    ('sweep_beta.py', ['10']),
    ('sweep_trigsums.py', ['10']),
    ('sweep_lightcurve.py', ['6']),
    ('sweep_timeseries.py', ['10']),
    ('sweep_profile_bkg.py', ['8']),
]


@pytest.mark.parametrize('script,extra', SWEEPS, ids=[s[0][:-3] for s in SWEEPS])
def test_sweep_finds_nothing(script, extra, capsys, monkeypatch):
    monkeypatch.setattr(sys, 'argv', [script, '7'] + extra)
    runpy.run_path(os.path.join(HERE, 'sweeps', script), run_name='__main__')
    out = capsys.readouterr().out
    last = [ln for ln in out.splitlines() if ln.strip()][-1]
    assert last == 'TOTAL BAD 0', out[-3000:]
