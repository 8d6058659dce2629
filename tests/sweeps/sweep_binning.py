"""Randomised comparison of Data.get_cells (cells and time series) with the oracle's bin_cells."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from godot_amd import core
from oracle import godot_oracle as orc
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
def n_(t): return t.cpu().numpy()
def same(a, b): 
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | ((a != a) & (b != b))).all())
nbad = 0
T0 = 2.4e8
for conf in range(nconf):
    rng = np.random.default_rng(7000 * seed0 + conf)
    nrow = int(rng.choice([50, 300, 2000, 9000, 40000]))
    # irregular spacecraft rows: variable lengths, gaps, some with zero live time
    lens = rng.choice([30., 30., 30., 15.5, 60., 7.25], nrow)
    gaps = np.where(rng.random(nrow) < 0.08, rng.choice([30., 900., 2400., 86400.], nrow), 0.0)
    TSTART = T0 + np.cumsum(gaps) + np.append(0, np.cumsum(lens[:-1]))
    TSTOP = TSTART + lens
    LIVETIME = lens * rng.uniform(0.7, 0.95, nrow)
    LIVETIME[rng.random(nrow) < 0.02] = 0.0
    exposure = LIVETIME * (3000 + 2000 * np.sin(2 * np.pi * (TSTART - T0) / 5700) ** 2)
    exposure[rng.random(nrow) < 0.05] = 0.0
    span = TSTOP[-1] - TSTART[0]
    nph = int(rng.choice([0, 5, 1000, 50000, 400000]))
    tcell = float(rng.choice([span / 7.3, 600., 100., 86400., 3000., span * 2, 31.])) 
    if span / tcell > 3e6: tcell = span / 1e5
    ti = TSTART[0] + span * rng.random(nph)
    if rng.random() < 0.5 and nph:
        # many photons exactly on cell edges and on row boundaries
        k = rng.random(nph) < 0.3
        ti[k] = TSTART[0] + np.round((ti[k] - TSTART[0]) / tcell) * tcell
        k = rng.random(nph) < 0.1
        ti[k] = TSTART[rng.integers(0, nrow, k.sum())]
    ti = np.sort(np.clip(ti, TSTART[0], TSTOP[-1]))
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    d = core.Data(ti, we, TSTART, TSTOP, LIVETIME, exposure)
    kw = {}
    if rng.random() < 0.5:
        kw['tstart'] = float(TSTART[0] + span * rng.uniform(-0.1, 0.5))
    if rng.random() < 0.5:
        kw['tstop'] = float(TSTOP[-1] - span * rng.uniform(-0.1, 0.4))
    trim = bool(rng.random() < 0.6)
    mfe = float(rng.choice([0, 0, 0.3]))
    what = []
    try:
        b = orc.bin_cells(ti, we, TSTART, TSTOP, exposure, LIVETIME, d.S, d.B, d.E, tcell=tcell,
                          trim_zero_exposure=trim, minimum_fractional_exposure=mfe, **kw)
        oerr = None
    except Exception as e:
        oerr = e
    try:
        cells = d.get_cells(tcell=tcell, use_barycenter=False, trim_zero_exposure=trim,
                            minimum_fractional_exposure=mfe, quiet=True, **kw)
        gerr = None
    except Exception as e:
        gerr = e
    if oerr is not None or gerr is not None:
        if (oerr is None) != (gerr is None):
            what.append('errors differ: oracle %r here %r' % (oerr, gerr))
    else:
        if not same(n_(cells.starts), b['starts']): what.append('starts')
        if not same(n_(cells.stops), b['stops']): what.append('stops')
        if len(cells) == len(b['counts']):
            if not np.array_equal(n_(cells.hi - cells.lo), b['counts']): what.append('counts')
            lo, hi = n_(cells.lo), n_(cells.hi)
            if len(lo) and nph:
                tt = np.concatenate([ti[a:c] for a, c in zip(lo, hi)]) if len(lo) < 200000 else None
                if tt is not None and not same(tt, b['times']): what.append('photons')
            if not same(n_(cells.sexp), b['sexp']): what.append('sexp %g' % np.abs(n_(cells.sexp) - b['sexp']).max())
            if not same(n_(cells.bexp), b['bexp']): what.append('bexp')
        else:
            what.append('ncell %d vs %d' % (len(cells), len(b['counts'])))
        # the time series of the same request
        ts = d.get_cells(tcell=tcell, use_barycenter=False, trim_zero_exposure=trim, time_series_only=True,
                         minimum_fractional_exposure=mfe, quiet=True, **kw)
        b = orc.zero_low_exposure(dict(b), 0)
        if not np.array_equal(n_(ts.counts), b['counts']): what.append('ts counts')
        if not same(n_(ts.exp), b['exp']): what.append('ts exp')
        if not np.allclose(n_(ts.weights), b['wsum'], rtol=1e-12, atol=1e-13): what.append('ts wsum')
        if not np.allclose(n_(ts.weights2), b['w2sum'], rtol=1e-12, atol=1e-13): what.append('ts w2sum')
    nbad += bool(what)
    print('conf %d rows %d nph %d tcell %g trim %d mfe %g kw %s -> %s cells: %s' % (
        conf, nrow, nph, tcell, trim, mfe, sorted(kw), 'err' if oerr is not None else len(b['counts']), what or 'ok'), flush=True)
print('TOTAL BAD', nbad)
