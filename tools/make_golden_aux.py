#!/usr/bin/env python3
"""Generate tests/golden/cell_aux.npz: the reference's small per-cell functions
(log_full_likelihood, fmin_tnc_func, f1/f2/f3, f1_profile,
log_profile_likelihood_approx, nr, halley, get_max_numerical,
get_beta_max_numerical, get_profile_flux) on a few cells.

Runs ONLY in the build container (needs /root/reference); see
make_golden.py for how the reference is imported.  Only data is written.

Usage:  python tools/make_golden_aux.py [outdir]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg   # noqa: E402

ALPHAS = [0.3, 1.0, 1.7, 4.2]
POINTS = [(0.4, 1.0), (1.0, 1.0), (1.3, 0.8), (2.5, 1.4)]


def main(outdir):
    from oracle import godot_oracle as orc
    core = mg.import_reference()
    d = orc.synth(60000, years=40 / 365.25, seed=37)
    data = mg.make_data(core, d)
    with mg.quiet():
        cells = data.get_cells(tcell=86400., use_barycenter=False)
        sparse = data.get_cells(tcell=1200., use_barycenter=False)
    rng = np.random.default_rng(1)
    pick = [cells[i] for i in (0, 7, 19)] + [sparse[int(i)] for i in rng.choice(len(sparse), 5, replace=False)]
    out = {}
    for ic, c in enumerate(pick):
        for swap in (False, True):
            tag = 'c%d%s_' % (ic, 's' if swap else '')
            out[tag + 'we'] = c.we.copy()
            out[tag + 'cell'] = np.array([c.tstart, c.tstop, c.exp, c.SonB])
            with mg.quiet():
                cl = core.CellLogLikelihood(c, swap=swap)
                out[tag + 'full'] = np.array([cl.log_full_likelihood(list(p)) for p in POINTS])
                tn = [cl.fmin_tnc_func(list(p)) for p in POINTS]
                out[tag + 'tnc'] = np.array([[t[0], t[1][0], t[1][1]] for t in tn])
                out[tag + 'f123'] = np.array([[cl.f1(a), cl.f2(a), cl.f3(a)] for a in ALPHAS])
                out[tag + 'f1_profile'] = np.array([cl.f1_profile(a) for a in ALPHAS])
                out[tag + 'lpl_approx'] = np.array([cl.log_profile_likelihood_approx(a) for a in ALPHAS])
                out[tag + 'nr'] = np.array([cl.nr(), cl.nr(guess=2.5, niter=3)], dtype=float)
                out[tag + 'halley'] = np.array([cl.halley(), cl.halley(guess=0.4, niter=2)], dtype=float)
                out[tag + 'max_num'] = np.array([cl.get_max_numerical(), cl.get_max_numerical(guess=3, beta=1.2)],
                                                dtype=float)
                out[tag + 'max_num_profile'] = np.asarray(cl.get_max_numerical(profile_background=True), dtype=float)
                out[tag + 'beta_num'] = np.array([cl.get_beta_max_numerical(a) for a in (0.5, 1.0, 2.0)], dtype=float)
                out[tag + 'profile_flux'] = np.array(cl.get_profile_flux())
                if ic in (0, 4):
                    with np.errstate(all='ignore'):
                        ag, bg, rv = cl.get_likelihood_grid(amax=3, bmax=2, res=0.25)
                    out[tag + 'grid_a'], out[tag + 'grid_b'], out[tag + 'grid'] = ag, bg, rv
    # many cells at once, the flat-layout class
    with mg.quiet():
        old = core.CellsLogLikelihoodOld(cells[:25])
        avec = np.random.default_rng(2).uniform(0.2, 3.0, 25)
        out['old_alpha'] = avec
        out['old_ll'] = np.array(old.log_likelihood(avec))
    out['old_ncell'] = np.array(25)
    # zero_weight: drop three days of exposure and their photons, then bin
    d2 = mg.make_data(core, d)
    d2.cexposure = np.cumsum(d2.exposure)
    t0 = d['TSTART'][0]
    with mg.quiet():
        d2.zero_weight(t0 + 3.2 * 86400, t0 + 6.1 * 86400)
        zc = d2.get_cells(tcell=86400., use_barycenter=False)
    out['zw_range'] = np.array([t0 + 3.2 * 86400, t0 + 6.1 * 86400])
    out['zw_totals'] = np.array([d2.S, d2.B, d2.E, len(d2.ti), len(d2.TSTART)])
    out['zw_starts'] = np.array([c.tstart for c in zc])
    out['zw_exp'] = np.array([c.exp for c in zc])
    out['zw_counts'] = np.array([len(c.we) for c in zc], dtype=np.int64)
    out.update(mg.inputs_of(d))
    out['ncase'] = np.array(len(pick))
    out['alphas'] = np.array(ALPHAS)
    out['points'] = np.array(POINTS)
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, 'cell_aux.npz'), **out)
    print('wrote', os.path.join(outdir, 'cell_aux.npz'), len(out), 'arrays')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
