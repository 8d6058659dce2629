"""Randomised comparison of cell_fit / cell_logpdf with the oracle over cell-size and weight distributions."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from godot_amd import ops
from oracle import godot_oracle as orc
RTOL = 1e-9
dev = torch.device('cuda:0')
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 30
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
t_start = time.time()
nbad = 0
for conf in range(nconf):
    if time.time() - t_start > budget:
        break
    rng = np.random.default_rng(1000 * seed0 + conf)
    mu = float(rng.choice([0.3, 1, 3, 6, 10, 15, 23, 30, 45, 60, 90, 150, 300, 800, 3000, 4500, 7000]))
    cdist = rng.choice(['poisson', 'geometric', 'constant', 'bimodal'])
    wdist = rng.choice(['b0315', 'b22', 'uniform', 'near1', 'tiny', 'mixed'])
    ncell = int(min(20000, max(60, 600000 / mu)))
    if cdist == 'poisson':
        cnt = rng.poisson(mu, ncell)
    elif cdist == 'geometric':
        cnt = rng.geometric(1.0 / (mu + 1), ncell) - 1
    elif cdist == 'constant':
        cnt = np.full(ncell, int(round(mu)))
    else:
        cnt = np.where(rng.random(ncell) < 0.9, rng.poisson(mu * 0.5, ncell), rng.poisson(mu * 6, ncell))
    cnt = cnt.astype(np.int64)
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    n = int(off[-1])
    if wdist == 'b0315':
        we = rng.beta(0.3, 1.5, n)
    elif wdist == 'b22':
        we = rng.beta(2, 2, n)
    elif wdist == 'uniform':
        we = rng.random(n)
    elif wdist == 'near1':
        we = rng.beta(5, 0.5, n)
    elif wdist == 'tiny':
        we = 1e-4 * rng.random(n)
    else:
        we = np.where(rng.random(n) < 0.5, rng.beta(0.3, 1.5, n), rng.beta(5, 0.5, n))
    we = np.clip(we, 1e-7, 1 - 1e-7)
    ew = we.mean() if n else 0.5
    f = float(rng.choice([0.2, 1.0, 1.0, 3.0]))
    S = np.maximum(1e-3, mu * ew * f * rng.uniform(0.7, 1.3, ncell))
    B = np.maximum(1e-3, mu * (1 - ew) * rng.uniform(0.7, 1.3, ncell))
    swap = bool(rng.random() < 0.25)
    beta = float(rng.choice([1.0, 1.0, 1.7, 0.6]))
    guess = float(rng.choice([1.0, 1.0, 0.2, 5.0]))
    wd, od = torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev)
    Sd, Bd = torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev)
    kw = {}
    if swap:
        cr = ops.CellRanges.from_offsets(1 - wd, od, Bd, Sd)
        kw['swap'] = True
    else:
        cr = ops.CellRanges.from_offsets(wd, od, Sd, Bd)
    if beta != 1.0: kw['beta'] = beta
    if guess != 1.0: kw['guess'] = guess
    alpha, ts, flags = ops.cell_fit(cr, **kw)
    alpha, ts, flags = alpha.cpu().numpy(), ts.cpu().numpy(), flags.cpu().numpy()
    tol = RTOL * (100 if swap else 1)
    # every cell (up to a CPU budget)
    ncheck = min(ncell, int(2.5e5 / max(mu, 10)) + 200)
    pick = np.unique(np.concatenate([rng.choice(ncell, ncheck, replace=False), np.flatnonzero(flags & 0x3e)[:200],
                                     np.argsort(cnt)[-20:], np.argsort(cnt)[:5]]))
    bad = 0
    worst = 0.0
    with np.errstate(all='ignore'):
        for c in pick:
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            try:
                a = cl.get_max(guess=guess, beta=beta)
            except Exception as e:
                if not (flags[c] & 0x10):
                    bad += 1
                    print('  cell', c, 'n', cnt[c], 'oracle raised', type(e).__name__, 'flags', flags[c], 'alpha', alpha[c])
                continue
            t = cl.get_ts(a)
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            ea = abs(alpha[c] - a) / max(1, abs(a))
            et = abs(ts[c] - t) / max(sc, abs(t))
            if np.isnan(a) and np.isnan(alpha[c]):
                continue
            worst = max(worst, ea, et)
            if not (ea <= tol and et <= tol) or ((flags[c] & 1) != 0) != (a == 0):
                bad += 1
                if bad <= 5:
                    print('  cell', c, 'n', cnt[c], 'alpha', alpha[c], a, 'ts', ts[c], t, 'flags', flags[c], 'S', S[c])
    # profile grids on a sample
    gbad = 0
    if mu <= 1000:
        dom, cod, gfl = ops.cell_logpdf(cr, dlogl=20, npt=100, **({'swap': True} if swap else {}))
        dom, cod, gfl = dom.cpu().numpy(), cod.cpu().numpy(), gfl.cpu().numpy()
        with np.errstate(all='ignore'):
            for c in pick[:: max(1, len(pick) // 60)]:
                w = we[off[c]:off[c + 1]]
                cl = orc.CellLike(w, S[c], S[c] / B[c])
                try:
                    d, v = cl.get_logpdf(dlogl=20, npt=100)
                except Exception:
                    if not (gfl[c] & 0x10): gbad += 1; print('  grid cell', c, 'oracle raised, flags', gfl[c])
                    continue
                if np.isnan(d).any() or np.isnan(v).any():
                    continue
                sc = max(1.0, np.abs(np.log1p(-w)).sum())
                if not (np.abs(dom[c] - d).max() <= tol * max(1.0, np.abs(d).max()) and
                        np.abs(cod[c] - v).max() <= tol * max(sc, np.abs(v).max())):
                    gbad += 1
                    if gbad <= 3:
                        print('  grid cell', c, 'n', cnt[c], 'ddom', np.abs(dom[c] - d).max(), 'dcod', np.abs(cod[c] - v).max(), 'flags', gfl[c])
    nbad += bad + gbad
    print('conf %d mu %g %s %s ncell %d swap %d beta %g guess %g f %g: checked %d, bad %d, grid bad %d, worst %.2e, flags %s  [%.0f s]' % (
        conf, mu, cdist, wdist, ncell, swap, beta, guess, f, len(pick), bad, gbad, worst,
        dict(zip(*np.unique(flags & 0x31, return_counts=True))), time.time() - t_start), flush=True)
print('TOTAL BAD', nbad)
