"""Randomised end-to-end comparison: Data.get_cells -> CellsLogLikelihood -> get_raw_lightcurve and
get_bb_lightcurve against the oracle's per-cell pipeline (bin_cells, get_logpdf, flux quantiles, Bayesian Blocks)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from godot_amd import core
from oracle import godot_oracle as orc
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 12
RTOL = 1e-8
nbad = 0
T0 = 2.4e8
t_begin = time.time()
for conf in range(nconf):
    rng = np.random.default_rng(5200 * seed0 + conf)
    nrow = int(rng.choice([3000, 30000]))
    lens = np.full(nrow, 30.)
    gaps = np.where(rng.random(nrow) < 0.05, rng.choice([30., 900., 2400., 40000.], nrow), 0.0)
    TSTART = T0 + np.cumsum(gaps) + np.append(0, np.cumsum(lens[:-1]))
    TSTOP = TSTART + lens
    LIVETIME = lens * rng.uniform(0.7, 0.95, nrow)
    exposure = LIVETIME * (3000 + 2000 * np.sin(2 * np.pi * (TSTART - T0) / 5700) ** 2)
    span = TSTOP[-1] - TSTART[0]
    nph = int(rng.choice([3000, 30000, 200000]))
    # a source that flares in the middle third: photon rate and weights change there
    u = rng.random(nph)
    flare = rng.random(nph) < 0.4
    tt = np.where(flare, 1 / 3 + u / 3, u)
    ti = np.sort(TSTART[0] + span * tt)
    mid = (ti > TSTART[0] + span / 3) & (ti < TSTART[0] + 2 * span / 3)
    we = np.clip(np.where(mid & (rng.random(nph) < 0.5), rng.beta(3, 1, nph), rng.beta(0.3, 1.5, nph)), 1e-6, 1 - 1e-6)
    ncell_goal = int(rng.choice([20, 150, 600]))
    tcell = span / (ncell_goal + 0.37)
    d = core.Data(ti, we, TSTART, TSTOP, LIVETIME, exposure)
    cells = d.get_cells(tcell=tcell, use_barycenter=False, quiet=True)
    b = orc.bin_cells(ti, we, TSTART, TSTOP, exposure, LIVETIME, d.S, d.B, d.E, tcell=tcell)
    clls = orc.cells_from_bins(b)
    what = []
    if len(cells) != len(clls):
        what.append('ncell')
    else:
        oerr = gerr = None
        try:
            with np.errstate(all='ignore'):
                dom, cod = orc.cells_logpdf(clls)
                ref = orc.raw_lightcurve(clls, b['starts'], b['stops'], dom, cod, tsmin=4)
        except (ValueError, RuntimeError) as e:
            oerr = e        # the reference itself stops here (no bracket for its root finder, endless re-centring)
        try:
            cll = core.CellsLogLikelihood(cells)
            lc = cll.get_raw_lightcurve(tsmin=4)
        except (ValueError, RuntimeError) as e:
            gerr = e
        if oerr is not None or gerr is not None:
            same = oerr is not None and gerr is not None and (isinstance(gerr, ValueError) == isinstance(oerr, ValueError))
            print('conf %d: oracle raised %r, here %r: %s' % (conf, oerr, gerr, 'same' if same else 'DIFFERENT'), flush=True)
            nbad += not same
            if not same or True:
                # what the device path makes of it when told to go on
                cll = core.CellsLogLikelihood(cells, allow_failed_cells=True)
                lc = cll.get_raw_lightcurve(tsmin=4)
                print('   with allow_failed_cells: %d of %d rows NaN' % (int(np.isnan(lc[:, 2]).sum()), len(lc)), flush=True)
            continue
        sc = np.array([max(1.0, np.abs(np.log1p(-c.we)).sum()) for c in clls])
        if not np.array_equal(lc[:, 0], ref[:, 0]): what.append('mjd')
        if not np.array_equal(lc[:, 4] == -1, ref[:, 4] == -1): what.append('UL branch')
        for k, scale in ((1, 1.0), (2, 1.0), (3, 1.0), (4, 1.0), (5, sc)):
            e = np.abs(lc[:, k] - ref[:, k]) / np.maximum(np.maximum(np.abs(ref[:, k]), 1.0), scale)
            fin = np.isfinite(ref[:, k])
            if not np.array_equal(np.isfinite(lc[:, k]), fin) or (fin.any() and e[fin].max() > RTOL):
                what.append('col %d err %.2e' % (k, np.nanmax(e)))
        # Bayesian Blocks on the same profiles
        tup = [(b['starts'][i], b['stops'][i], b['sexp'][i], b['times'][b['offsets'][i]:b['offsets'][i + 1]],
                b['weights'][b['offsets'][i]:b['offsets'][i + 1]], b['sexp'][i] / b['bexp'][i]) for i in range(len(clls))]
        prior = float(rng.choice([2, 8]))
        try:
            with np.errstate(all='ignore'):
                rrows, ridx = orc.bb_lightcurve(tup, dom, cod, tsmin=4, bb_prior=prior)
        except (ValueError, RuntimeError) as e:
            print('conf %d: the oracle raised %s in the block light curve: skipped' % (conf, type(e).__name__), flush=True)
            continue
        rows, idx = cll.get_bb_lightcurve(tsmin=4, bb_prior=prior, get_indices=True)
        if not np.array_equal(idx, ridx):
            what.append('bb blocks %d vs %d' % (len(idx) - 1, len(ridx) - 1))
        else:
            if not np.array_equal(rows[:, 4] == -1, rrows[:, 4] == -1): what.append('bb UL branch')
            e = np.abs(rows - rrows) / np.maximum(np.abs(rrows), 1.0)
            e[:, 5] = np.abs(rows[:, 5] - rrows[:, 5]) / np.maximum(np.abs(rrows[:, 5]), 100.0)
            if np.nanmax(e) > 1e-7: what.append('bb rows err %.2e in col %d' % (np.nanmax(e), np.nanargmax(e) % 6))
    nbad += bool(what)
    print('conf %d rows %d nph %d ncell %d blocks %s: %s [%.0f s]' % (conf, nrow, nph, len(clls), (len(idx) - 1) if not what or 'ncell' not in what else '-', what or 'ok', time.time() - t_begin), flush=True)
print('TOTAL BAD', nbad)
