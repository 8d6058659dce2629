#!/usr/bin/env python3
"""Generate tests/golden/bayesian_blocks.npz by running the reference's
Bayesian-Blocks code (CellsLogLikelihood.do_bb / get_bb_lightcurve /
do_top_hat) on a synthetic source with two flares.

Runs ONLY in the build container (needs /root/reference); see
make_golden.py for how the reference is imported.  Only data is written.

Usage:  python tools/make_golden_bb.py [outdir]
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
    d = orc.synth(60000, years=30 / 365.25, seed=19)
    # two flares: photons inside become more source-like (weights pushed towards 1)
    t = d['ti'] - d['ti'][0]
    for (a, b, f) in ((6.0, 8.5, 0.25), (19.0, 19.8, 0.1)):
        m = (t > a * 86400) & (t < b * 86400)
        d['we'] = np.where(m, 1 - (1 - d['we']) * f, d['we'])
    d['we'] = np.clip(d['we'], 1e-6, 1 - 1e-6)
    data = mg.make_data(core, d)
    with mg.quiet():
        cells = data.get_cells(tcell=7200., use_barycenter=False)
        cll = core.CellsLogLikelihood(cells)
        rows, idx = cll.get_bb_lightcurve(tsmin=4, bb_prior=8, get_indices=True)
        fv = cll.fitness_vals
        fit_val = np.concatenate([x[0] for x in fv])
        fit_arg = np.concatenate([x[1] for x in fv])
        bb8 = cll.do_bb(prior=8)            # the cached-rows path
        bb2 = cll.do_bb(prior=2)
        top = cll.do_top_hat()
        sl = cll.fitness(5, 40)
    n = len(cells)
    out = dict(
        starts=np.array([c.tstart for c in cells]), stops=np.array([c.tstop for c in cells]),
        exp=np.array([c.exp for c in cells]), SonB=np.array([c.SonB for c in cells]),
        counts=np.array([len(c.we) for c in cells], dtype=np.int64),
        we=np.concatenate([c.we for c in cells]), ti=np.concatenate([c.ti for c in cells]),
        dom=cll._dom, cod=cll._cod, fit_val=fit_val, fit_arg=fit_arg,
        bb_rows=rows, bb_idx=np.asarray(idx, dtype=np.int64),
        bb8_indices=np.asarray(bb8[0], dtype=np.int64), bb8_scalars=np.array([bb8[1], bb8[2], bb8[3]], dtype=float),
        bb2_indices=np.asarray(bb2[0], dtype=np.int64), bb2_scalars=np.array([bb2[1], bb2[2], bb2[3]], dtype=float),
        top_hat_diag=np.stack([np.stack([np.diagonal(top[k], offset=o)[: n - 7] for o in (0, 1, 7)])
                               for k in (0, 1)]),
        fitness_5_40=sl)
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, 'bayesian_blocks.npz'), **out)
    print('bayesian_blocks: %d cells, %d photons, %d blocks (prior 8), %d blocks (prior 2)' % (
        n, len(out['we']), len(bb8[0]), len(bb2[0])))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(mg.REPO, 'tests', 'golden'))
