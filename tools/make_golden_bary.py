#!/usr/bin/env python3
"""Generate tests/golden/bary_cells.npz by running the reference's
get_cells(use_barycenter=True) with analytic time converters in place of the
ephemeris-based interpolators (the reference caches its converters in the
attributes _t2b / _b2t; setting them beforehand is all it takes).

Runs ONLY in the build container (needs /root/reference); see
make_golden.py for how the reference is imported.  Only data is written.

Usage:  python tools/make_golden_bary.py [outdir]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg   # noqa: E402
from make_golden_variants import per_cell   # noqa: E402

AMP, PERIOD = 400.0, 3.15576e7   # a Roemer-like delay: 400 s amplitude, one year


def converters(t0):
    """topo -> bary adds a sinusoidal delay; bary -> topo inverts it by fixed-point iteration."""
    def t2b(t):
        t = np.asarray(t, dtype=np.float64)
        return t + AMP * np.sin(2 * np.pi * (t - t0) / PERIOD)

    def b2t(b):
        b = np.asarray(b, dtype=np.float64)
        t = b.copy()
        for _ in range(6):
            t = b - AMP * np.sin(2 * np.pi * (t - t0) / PERIOD)
        return t
    return t2b, b2t


def main(outdir):
    from oracle import godot_oracle as orc
    core = mg.import_reference()
    d = orc.synth(30000, years=12 / 365.25, seed=29)
    t0 = d['TSTART'][0]
    out = dict(mg.inputs_of(d))
    out.update(bary_amp=AMP, bary_period=PERIOD)
    for timeref, tag in (('LOCAL', 'topo'), ('SOLARSYSTEM', 'ssb')):
        data = mg.make_data(core, d)
        data.timeref = timeref
        data._t2b, data._b2t = converters(t0)
        with mg.quiet():
            cells = data.get_cells(tcell=3600., use_barycenter=True)
            for k, v in per_cell(cells).items():
                out['%s_%s' % (tag, k)] = v
            out[tag + '_topo_edges'] = data._topo_edges.copy()
            ts = data.get_cells(tcell=1800., use_barycenter=True, time_series_only=True,
                                trim_zero_exposure=False)
            out.update({tag + '_ts_starts': ts.starts, tag + '_ts_stops': ts.stops, tag + '_ts_exp': ts.exp,
                        tag + '_ts_sexp': ts.sexp, tag + '_ts_bexp': ts.bexp,
                        tag + '_ts_counts': np.asarray(ts.counts, dtype=np.int64),
                        tag + '_ts_weights': ts.weights, tag + '_ts_weights2': ts.weights2,
                        tag + '_ts_alt_starts': ts.alt_starts, tag + '_ts_alt_stops': ts.alt_stops,
                        tag + '_ts_timesys': np.array(ts.timesys)})
            if tag == 'topo':
                # the interchange file between the binning and spectrum stages, as the
                # reference writes it (CellTimeSeries.save, core.py:246-260)
                os.makedirs(outdir, exist_ok=True)
                ts.save(os.path.join(outdir, 'cell_timeseries_ref.npz'))
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, 'bary_cells.npz'), **out)
    print('wrote', os.path.join(outdir, 'bary_cells.npz'), len(out), 'arrays')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
