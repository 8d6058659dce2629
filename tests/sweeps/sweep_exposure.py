"""Randomised comparison of Data.get_exposure / get_deadtime with the oracle's exposure_at on arbitrary
query times: inside rows, inside gaps, exactly on row starts and stops, before the table, sorted or not,
cumulative or differenced.  Every number must agree to the bit."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from godot_amd import core
from oracle import godot_oracle as orc
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
def n_(t): return None if t is None else t.cpu().numpy()
def same(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and bool(((a == b) | ((a != a) & (b != b))).all())
nbad = 0
T0 = 2.4e8
for conf in range(nconf):
    rng = np.random.default_rng(8300 * seed0 + conf)
    # short tables take the serial running sums, long ones the parallel scan
    nrow = int(rng.choice([1, 2, 50, 3000, 8191, 8192, 8193, 50000, 300000]))
    lens = rng.choice([30., 30., 30., 15.5, 60., 7.25], nrow)
    gaps = np.where(rng.random(nrow) < 0.08, rng.choice([30., 900., 2400., 86400.], nrow), 0.0)
    TSTART = T0 + np.cumsum(gaps) + np.append(0, np.cumsum(lens[:-1]))
    TSTOP = TSTART + lens
    LIVETIME = lens * rng.uniform(0.7, 0.95, nrow)
    LIVETIME[rng.random(nrow) < 0.02] = 0.0
    scale = float(rng.choice([1.0, 1e-6, 1e7]))
    exposure = scale * LIVETIME * (3000 + 2000 * np.sin(2 * np.pi * (TSTART - T0) / 5700) ** 2)
    exposure[rng.random(nrow) < 0.05] = 0.0
    if rng.random() < 0.2:
        # long stretches without exposure
        k = rng.integers(0, nrow)
        exposure[k:k + nrow // 3] = 0.0
    if exposure.sum() == 0:
        exposure[0] = 1.0
    span = TSTOP[-1] - TSTART[0]
    nph = 1000
    ti = np.sort(TSTART[0] + span * rng.random(nph))
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    d = core.Data(ti, we, TSTART, TSTOP, LIVETIME, exposure)
    nq = int(rng.choice([1, 2, 17, 1000, 100000]))
    q = TSTART[0] + span * rng.random(nq)
    k = rng.random(nq) < 0.2
    q[k] = TSTART[rng.integers(0, nrow, k.sum())]
    k = rng.random(nq) < 0.2
    q[k] = TSTOP[rng.integers(0, nrow, k.sum())]
    k = rng.random(nq) < 0.05
    q[k] = TSTART[0] - rng.random(k.sum()) * 100          # before the table: clipped to its first row
    k = rng.random(nq) < 0.1
    q[k] = np.nextafter(TSTOP[rng.integers(0, nrow, k.sum())], -np.inf)
    q = np.minimum(q, TSTOP[-1])
    if rng.random() < 0.7:
        q = np.sort(q)
    dead = bool(rng.random() < 0.5)
    cum = bool(rng.random() < 0.5)
    what = []
    ref = orc.exposure_at(TSTART, TSTOP, exposure, LIVETIME, d.S, d.B, d.E, q, deadtime_too=dead, cumulative=cum)
    got = d.get_exposure(q, deadtime_too=dead, cumulative=cum)
    for name, a, b in zip(('exp', 'sexp', 'bexp', 'dead'), got, ref):
        if (a is None) != (b is None):
            what.append(name + ' missing')
        elif a is not None and not same(n_(a), b):
            what.append('%s %g' % (name, np.nanmax(np.abs(n_(a) - b) / np.maximum(np.abs(b), 1e-300))))
    if not same(n_(d.get_deadtime(q)), orc.exposure_at(TSTART, TSTOP, exposure, LIVETIME, d.S, d.B, d.E, q,
                                                       deadtime_too=True)[3]):
        what.append('get_deadtime')
    nbad += bool(what)
    print('conf %d rows %d queries %d scale %g dead %d cumulative %d: %s' % (conf, nrow, nq, scale, dead, cum,
                                                                         what or 'ok'), flush=True)
print('TOTAL BAD', nbad)
