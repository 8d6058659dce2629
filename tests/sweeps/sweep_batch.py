"""Randomised comparison of the batched binning of many sources (get_cells_batch,
get_timeseries_batch: two launches for all sources) with the same sources binned one at a
time, bit for bit; irregular spacecraft tables, spans that clip differently per source."""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from godot_amd import core
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 30
T0 = 2.4e8


def source(rng):
    nrow = int(rng.choice([50, 300, 2000, 9000]))
    lens = rng.choice([30., 30., 30., 15.5, 60., 7.25], nrow)
    gaps = np.where(rng.random(nrow) < 0.08, rng.choice([30., 900., 2400., 86400.], nrow), 0.0)
    TSTART = T0 + float(rng.choice([0., 0., 1234.5])) + np.cumsum(gaps) + np.append(0, np.cumsum(lens[:-1]))
    TSTOP = TSTART + lens
    LIVETIME = lens * rng.uniform(0.7, 0.95, nrow)
    LIVETIME[rng.random(nrow) < 0.02] = 0.0
    exposure = LIVETIME * (3000 + 2000 * np.sin(2 * np.pi * (TSTART - T0) / 5700) ** 2)
    exposure[rng.random(nrow) < 0.05] = 0.0
    span = TSTOP[-1] - TSTART[0]
    nph = int(rng.choice([1, 5, 1000, 50000]))
    ti = np.sort(np.clip(TSTART[0] + span * rng.random(nph), TSTART[0], TSTOP[-1]))
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    return core.Data(ti, we, TSTART, TSTOP, LIVETIME, exposure), span


def same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a.view(torch.int64), b.view(torch.int64)))


nbad = 0
worst = 0.0
for conf in range(nconf):
    rng = np.random.default_rng(9100 * seed0 + conf)
    made = [source(rng) for _ in range(int(rng.integers(2, 7)))]
    datas = [d for d, _ in made]
    span = min(s for _, s in made)
    tcell = float(rng.choice([span / 7.3, 600., 100., 86400., 3000., span * 2, 31.]))
    if max(s for _, s in made) / tcell > 2e6:
        tcell = max(s for _, s in made) / 1e5
    kw = {}
    if rng.random() < 0.5:
        kw['tstart'] = float(T0 + span * rng.uniform(-0.1, 0.5))
    if rng.random() < 0.5:
        kw['tstop'] = float(T0 + span * rng.uniform(0.6, 1.3))
    trim = bool(rng.random() < 0.6)
    if rng.random() < 0.5:
        core.pack_sources(datas)
    what = []
    try:
        each = core.get_cells_batch(datas, tcell=tcell, trim_zero_exposure=trim, per_source_calls=True, **kw)
        eerr = None
    except Exception as e:
        eerr = e
    try:
        one = core.get_cells_batch(datas, tcell=tcell, trim_zero_exposure=trim, **kw)
        oerr = None
    except Exception as e:
        oerr = e
    if eerr is not None or oerr is not None:
        if type(eerr) is not type(oerr):
            what.append('errors differ: per source %r batch %r' % (eerr, oerr))
    else:
        if not np.array_equal(one.cell_offsets, each.cell_offsets): what.append('offsets')
        if not torch.equal(one.source, each.source): what.append('source')
        for nm in ('lo', 'hi', 'starts', 'stops', 'sexp', 'bexp', 'we', 'ti'):
            if not same(getattr(one.cells, nm), getattr(each.cells, nm)): what.append(nm)
        got = core.get_timeseries_batch(datas, tcell=tcell, trim_zero_exposure=trim, **kw)
        for k, (d, ts) in enumerate(zip(datas, got)):
            ref = d.get_cells(tcell=tcell, time_series_only=True, use_barycenter=False, trim_zero_exposure=trim,
                              quiet=True, **kw)
            for nm in ('starts', 'stops', 'exp', 'sexp', 'bexp', 'deadtime', 'counts'):
                if not same(getattr(ts, nm), getattr(ref, nm)): what.append('ts %d %s' % (k, nm))
            for nm in ('weights', 'weights2'):
                a, b = getattr(ts, nm).cpu().numpy(), getattr(ref, nm).cpu().numpy()
                # sums of positive terms: a cell's photons may be added in another order when the
                # cells of all sources share a launch (the kernel's shape follows the mean cell size)
                if a.shape != b.shape:
                    what.append('ts %d %s shape' % (k, nm))
                elif a.size and not np.all(np.abs(a - b) <= 1e-13 * np.maximum(np.abs(b), 1.0)):
                    what.append('ts %d %s %.3g' % (k, nm, np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))))
                elif a.size:
                    worst = max(worst, float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))))
    nbad += bool(what)
    print('conf %d sources %d tcell %g trim %d kw %s -> %s: %s' % (
        conf, len(datas), tcell, trim, sorted(kw), 'err %r' % oerr if oerr is not None else '%d cells' % len(one.cells.lo),
        what or 'ok'), flush=True)
print('largest relative difference of a weight moment: %.3g' % worst)
print('TOTAL BAD', nbad)
