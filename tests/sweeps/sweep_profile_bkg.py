"""Randomised check of the background-profiled path: cell_fit_joint against the oracle's fmin_tnc fit (contract:
logL not lower than the reference's by more than its own ftol, vanishing projected gradient, logL reproduced),
cell_logpdf_profile against the oracle's grid for the same centre."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from godot_amd import ops
from oracle import godot_oracle as orc
dev = torch.device('cuda:0')
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nbad = 0
t0 = time.time()
for conf in range(nconf):
    rng = np.random.default_rng(6100 * seed0 + conf)
    mu = float(rng.choice([2, 6, 15, 30, 80, 300]))
    ncell = int(rng.choice([100, 300]))
    cnt = np.maximum(rng.poisson(mu, ncell), 0).astype(np.int64)
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    n = int(off[-1])
    wl = rng.choice(['b0315', 'b22', 'near1', 'mixed'])
    we = {'b0315': rng.beta(0.3, 1.5, n), 'b22': rng.beta(2, 2, n), 'near1': rng.beta(5, 0.5, n),
          'mixed': np.where(rng.random(n) < 0.5, rng.beta(0.3, 1.5, n), rng.beta(5, 0.5, n))}[wl]
    we = np.clip(we, 1e-6, 1 - 1e-6)
    ew = we.mean() if n else 0.5
    S = np.maximum(1e-2, mu * ew * rng.choice([0.3, 1, 3]) * rng.uniform(0.7, 1.3, ncell))
    B = np.maximum(1e-2, mu * (1 - ew) * rng.choice([0.3, 1, 3]) * rng.uniform(0.7, 1.3, ncell))
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    a, b, L, fl = [x.cpu().numpy() for x in ops.cell_fit_joint(cr)]
    bad = 0
    worst_gap = 0.0
    nfail = int(((fl & 0x10) != 0).sum())
    refs = np.full((ncell, 2), np.nan)
    with np.errstate(all='ignore'):
        for i in range(ncell):
            w = we[off[i]:off[i + 1]]
            cl = orc.CellLike(w, S[i], S[i] / B[i])
            try:
                ra, rb = cl.get_max_profile()
            except Exception as e:
                continue
            refs[i] = ra, rb
            Lr = cl.log_full_likelihood(ra, rb)
            if fl[i] & 0x10:
                bad += 1
                if bad < 4: print('  cell', i, 'n', cnt[i], 'flagged failed; reference', ra, rb)
                continue
            gap = Lr - L[i]
            worst_gap = max(worst_gap, gap)
            _, (nga, ngb) = cl.neg_logl_grad(a[i], b[i])
            sc = max(1.0, cl.S, cl.B)
            okg = ((abs(nga) <= 1e-7 * sc) or (a[i] == 0 and nga >= 0)) and ((abs(ngb) <= 1e-7 * sc) or (b[i] == 0 and ngb >= 0))
            okl = abs(L[i] - cl.log_full_likelihood(a[i], b[i])) <= 1e-9 * max(1, abs(L[i]))
            if not (gap <= 2e-3 and okg and okl and a[i] >= 0 and b[i] >= 0):
                bad += 1
                if bad < 4: print('  cell', i, 'n', cnt[i], 'S', S[i], 'B', B[i], 'here', a[i], b[i], L[i], 'ref', ra, rb, Lr, 'grad', nga, ngb)
    # profile grid for the reference's centres
    sel = np.flatnonzero(np.isfinite(refs[:, 0]))[:40]
    gbad = 0
    if len(sel):
        sub = ops.CellRanges(cr.we, cr.lo[sel].contiguous(), cr.hi[sel].contiguous(), cr.S[sel].contiguous(), cr.B[sel].contiguous())
        dom, cod, gfl = [x.cpu().numpy() for x in ops.cell_logpdf_profile(sub, aopt=torch.from_numpy(refs[sel, 0].copy()).to(dev), dlogl=20, npt=30)]
        with np.errstate(all='ignore'):
            for k, i in enumerate(sel):
                w = we[off[i]:off[i + 1]]
                cl = orc.CellLike(w, S[i], S[i] / B[i])
                try:
                    d, v = cl.get_logpdf_profile(aopt=refs[i, 0], dlogl=20, npt=30)
                except Exception as e:
                    if not (gfl[k] & 0x10):
                        gbad += 1
                        if gbad < 4: print('  grid cell', i, 'oracle raised', type(e).__name__, 'flags', gfl[k])
                    continue
                if np.isnan(d).any() or np.isnan(v).any():
                    continue
                sc = max(1.0, np.abs(np.log1p(-w)).sum())
                ed = np.abs(dom[k] - d).max() / max(1.0, np.abs(d).max())
                ev = np.abs(cod[k] - v).max() / max(sc, np.abs(v).max())
                if not (ed <= 1e-7 and ev <= 1e-7):
                    gbad += 1
                    if gbad < 4: print('  grid cell', i, 'n', cnt[i], 'ddom %.2e dcod %.2e' % (ed, ev), 'flags', gfl[k], 'ends', dom[k][[0, -1]], d[[0, -1]])
    nbad += bad + gbad
    print('conf %d mu %g %s: fit bad %d (flagged %d), worst logL gap %.2e, grid bad %d of %d [%.0f s]' % (conf, mu, wl, bad, nfail, worst_gap, gbad, len(sel), time.time() - t0), flush=True)
print('TOTAL BAD', nbad)
