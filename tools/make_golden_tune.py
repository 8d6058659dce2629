#!/usr/bin/env python3
"""Generate tests/golden/bb_prior_tune.npz by running the reference's bb_prior_tune
(core.py:2784-2817) on synthetic data with a fixed seed.  Runs ONLY in the build
container (needs /root/reference).  Only data is written.

Usage:  python tools/make_golden_tune.py [outdir]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg   # noqa: E402


def main(outdir):
    from oracle import godot_oracle as orc
    core = mg.import_reference()
    d = orc.synth(20000, years=10 / 365.25, seed=29)
    data = mg.make_data(core, d)
    priors = [2, 4, 8]
    with mg.quiet():
        r1, n1 = core.bb_prior_tune(data, 7200., bb_priors=priors, ntrial=2, use_barycenter=False, seed=7)
        r2, n2 = core.bb_prior_tune(data, None, bb_priors=priors, ntrial=1, orbital=True, seed=8)
    out = dict(mg.inputs_of(d))
    out.update(priors=np.array(priors), tune_cells=r1, tune_ncell=n1, tune_orbital=r2, tune_orbital_ncell=n2)
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, 'bb_prior_tune.npz'), **out)
    print('bb_prior_tune:', r1.tolist(), n1, r2.tolist(), n2)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(mg.REPO, 'tests', 'golden'))
