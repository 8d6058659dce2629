"""Randomised comparison of rfft, Z^2_m / H and the Bayesian-Blocks kernels with numpy / the oracle."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from godot_amd import ops
from oracle import godot_oracle as orc
dev = torch.device('cuda:0')
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def d_(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
nbad = 0
# ---- rfft ------------------------------------------------------------------
worst = 0
for conf in range(60):
    rng = np.random.default_rng(9000 * seed0 + conf)
    kind = rng.choice(['pow2', 'small', 'odd', 'prime', 'smooth'])
    if kind == 'pow2': l = 1 << int(rng.integers(1, 22))
    elif kind == 'small': l = int(rng.integers(2, 200))
    elif kind == 'odd': l = int(rng.integers(3, 300000)) | 1
    elif kind == 'prime': l = int(rng.choice([3, 5, 7, 11, 13, 17, 257, 4099, 65537, 100003, 1048583]))
    else: l = int(2 ** rng.integers(0, 10) * 3 ** rng.integers(0, 6) * 5 ** rng.integers(0, 4))
    l = max(2, min(l, 1 << 21))
    n = int(rng.choice([l, l, max(1, l // 2), max(1, l - 1), l + 5]))
    x = rng.normal(0, 1, n) * rng.choice([1.0, 1e6, 1e-6]) + rng.choice([0.0, 100.0])
    try:
        out = ops.rfft(d_(x), l).cpu().numpy()
        err = None
    except Exception as e:
        err = e
    ref = np.fft.rfft(x, n=l)
    if err is not None:
        print('rfft conf %d l %d n %d: raised %r' % (conf, l, n, err)); nbad += 1; continue
    got = out[:, 0] + 1j * out[:, 1]
    e = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)
    worst = max(worst, e)
    if not e < 1e-11:
        nbad += 1
        print('rfft conf %d %s l %d n %d: rel err %.2e' % (conf, kind, l, n, e))
print('rfft worst %.2e' % worst, flush=True)
# ---- Z^2_m / H ---------------------------------------------------------------
worst = 0
for conf in range(40):
    rng = np.random.default_rng(9500 * seed0 + conf)
    nph = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 30000]))
    m = int(rng.choice([1, 2, 2, 3, 4, 5, 7, 10, 16, 20]))
    nf = int(rng.choice([1, 2, 19, 20, 21, 47, 64, 500, 1283]))
    f0 = float(rng.choice([0.0, 1e-6, 0.37, 29.6, 641.9]))
    df = float(rng.choice([1e-9, 3.17e-8, 1e-5, 0.01]))
    j0 = int(rng.choice([0, 0, 5, 1 << 20, 1 << 33]))
    t = np.sort(rng.uniform(2.4e8, 5.6e8, nph))
    w = rng.random(nph)
    sign = rng.random() < 0.25
    if sign: w[rng.random(nph) < 0.3] *= -1
    if rng.random() < 0.2: w[rng.random(nph) < 0.5] = 0.0
    if not (w != 0).any(): w[0] = 0.5
    want_h = bool(rng.random() < 0.5)
    z, h = ops.z2m(d_(t), d_(w), f0, df, nf, m=m, j0=j0, want_h=want_h)
    z = z.cpu().numpy()
    zr = orc.z2m(t, w, f0, df, nf, m=m, j0=j0)
    sc = max(1.0, np.abs(zr).max())
    # absolute phase error grows as J^2/2 ulp times 2 pi m; scale the bound by the sum of |w|
    bound = 1e-9 * max(sc, 2 * np.abs(w).sum() ** 2 / (w ** 2).sum())
    e = np.abs(z - zr).max()
    worst = max(worst, e / bound)
    bad = not e <= bound
    if want_h:
        hr = orc.h_test(t, w, f0, df, nf, mmax=m, j0=j0)
        eh = np.abs(h.cpu().numpy() - hr).max()
        bad = bad or not eh <= bound
    if bad:
        nbad += 1
        print('z2m conf %d nph %d m %d nf %d f0 %g df %g j0 %d sign %d: err %.2e bound %.2e' % (conf, nph, m, nf, f0, df, j0, sign, e, bound))
print('z2m worst err/bound %.2e' % worst, flush=True)
# ---- Bayesian Blocks: fitness rows and partition -----------------------------
for conf in range(12):
    rng = np.random.default_rng(9900 * seed0 + conf)
    ncell = int(rng.choice([1, 2, 5, 33, 64, 65, 200, 700]))
    npt = int(rng.choice([2, 3, 50, 200, 201]))
    ngrid = int(rng.choice([1000, 1000, 100, 1024, 7]))
    dom = np.empty((ncell, npt)); cod = np.empty((ncell, npt))
    for c in range(ncell):
        a0 = rng.uniform(0, 2) if rng.random() < 0.7 else 0.0
        a1 = a0 + rng.uniform(0.05, 3)
        dom[c] = np.linspace(a0, a1, npt)
        if rng.random() < 0.15:   # not a linspace
            dom[c] = np.sort(rng.uniform(a0, a1, npt)); dom[c, 0] = a0; dom[c, -1] = a1
        pk = rng.uniform(a0, a1)
        cod[c] = -rng.uniform(1, 50) * (dom[c] - pk) ** 2
        if rng.random() < 0.2: cod[c, 0] = -np.inf
        if rng.random() < 0.1: cod[c, rng.integers(0, npt)] = -np.inf
    if ngrid != 1000:
        continue   # the oracle's grid is fixed at 1000 points
    fv, fa = ops.bb_fitness(d_(dom), d_(cod), ngrid=ngrid)
    fv, fa = fv.cpu().numpy(), fa.cpu().numpy()
    bad = 0
    with np.errstate(all='ignore'):
        for i in range(ncell):
            rv, ra = orc.bb_fitness(dom, cod, 0, i + 1)[0:2]
            o = i * (i + 1) // 2
            if not (np.array_equal(fv[o:o + i + 1], rv, equal_nan=True) and np.array_equal(fa[o:o + i + 1], ra, equal_nan=True)):
                bad += 1
        rows = [fv[i * (i + 1) // 2:(i + 1) * (i + 2) // 2] for i in range(ncell)]
        prior = float(rng.choice([0.5, 2, 8]))
        idx, bl, best_all, last_ref = orc.bb_partition(rows, prior)
    best, last = ops.bb_partition(d_(fv), ncell, prior)
    if not (np.array_equal(last.cpu().numpy(), last_ref) and np.array_equal(best.cpu().numpy(), best_all, equal_nan=True)):
        bad += 1
    nbad += bool(bad)
    print('bb conf %d ncell %d npt %d: %s' % (conf, ncell, npt, 'ok' if not bad else 'BAD %d' % bad), flush=True)
print('TOTAL BAD', nbad)
