#!/usr/bin/env python3
"""Generate tests/golden/phase_cells.npz by running the reference's PhaseData.get_cells
and PhaseCellsLogLikelihood (Bayesian Blocks on a folded light curve) on synthetic
pulse phases.  Runs ONLY in the build container (needs /root/reference); see
make_golden.py for how the reference is imported.  Only data is written.

Usage:  python tools/make_golden_phase.py [outdir]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg   # noqa: E402
from make_golden_variants import per_cell   # noqa: E402


def main(outdir):
    core = mg.import_reference()
    rng = np.random.default_rng(41)
    n = 20000
    we = np.clip(rng.beta(0.3, 1.5, n), 1e-6, 1 - 1e-6)
    # a pulse: source-like photons cluster around phase 0.3
    ph = np.where(rng.random(n) < we * 0.6, np.mod(0.3 + 0.04 * rng.standard_normal(n), 1), rng.random(n))
    a = np.argsort(ph)
    ph, we = ph[a], we[a]
    pd = core.PhaseData.__new__(core.PhaseData)
    pd.ti, pd.we = ph.copy(), we.copy()
    pd.E = 1
    pd.S = np.sum(pd.we)
    pd.B = len(pd.we) - pd.S
    out = dict(phases=ph, weights=we)
    with mg.quiet():
        cells = pd.get_cells(ncell=50)
        for k, v in per_cell(cells).items():
            out['plain_' + k] = v
        cells_c = pd.get_cells(ncell=20, get_converse=True)
        for k, v in per_cell(cells_c).items():
            out['converse_' + k] = v
        cells_r = pd.get_cells(ncell=50, randomize=True, seed=9)
        for k, v in per_cell(cells_r).items():
            out['rand_' + k] = v
        pcl = core.PhaseCellsLogLikelihood(cells)
        rows, idx = pcl.get_bb_lightcurve(tsmin=4, bb_prior=8, get_indices=True)
        out.update(bb_rows=rows, bb_idx=np.asarray(idx, dtype=np.int64))
        raw = core.CellsLogLikelihood(cells).get_raw_lightcurve()
        out['raw_rows'] = raw
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, 'phase_cells.npz'), **out)
    print('phase_cells: %d photons, %d blocks over three wraps' % (n, len(rows)))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(mg.REPO, 'tests', 'golden'))
