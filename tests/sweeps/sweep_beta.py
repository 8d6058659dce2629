"""Randomised comparison of cell_beta_max / log_profile_likelihood / cell_loglike / cell_sums with the oracle."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from godot_amd import ops
from oracle import godot_oracle as orc
RTOL = 1e-9
dev = torch.device('cuda:0')
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nbad = 0
t0 = time.time()
for conf in range(nconf):
    rng = np.random.default_rng(4000 * seed0 + conf)
    mu = float(rng.choice([1, 3, 10, 23, 45, 90, 300, 2000]))
    ncell = int(min(4000, max(40, 200000 / mu)))
    cdist = rng.choice(['poisson', 'geometric'])
    cnt = (rng.poisson(mu, ncell) if cdist == 'poisson' else rng.geometric(1 / (mu + 1), ncell) - 1).astype(np.int64)
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    n = int(off[-1])
    wd_ = rng.choice(['b0315', 'b22', 'near1', 'mixed'])
    we = {'b0315': rng.beta(0.3, 1.5, n), 'b22': rng.beta(2, 2, n), 'near1': rng.beta(5, 0.5, n),
          'mixed': np.where(rng.random(n) < 0.5, rng.beta(0.3, 1.5, n), rng.beta(5, 0.5, n))}[wd_]
    we = np.clip(we, 1e-7, 1 - 1e-7)
    ew = we.mean()
    S = np.maximum(1e-3, mu * ew * rng.choice([0.3, 1, 3]) * rng.uniform(0.7, 1.3, ncell))
    B = np.maximum(1e-3, mu * (1 - ew) * rng.choice([0.3, 1, 3]) * rng.uniform(0.7, 1.3, ncell))
    alpha = rng.choice([0.0, 0.3, 1.0, 2.5, 10.0], ncell) * rng.uniform(0.8, 1.2, ncell)
    guess = float(rng.choice([1.0, 1.0, 0.3, 4.0]))
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    ad = torch.from_numpy(alpha).to(dev)
    beta, lp, fl = ops.cell_beta_max(cr, ad, guess=guess)
    ll = ops.cell_loglike(cr, ad).cpu().numpy()
    beta, lp, fl = beta.cpu().numpy(), lp.cpu().numpy(), fl.cpu().numpy()
    bad = 0
    worst = 0
    nerr = 0
    with np.errstate(all='ignore'):
        for c in range(ncell):
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            lr = cl.log_likelihood(alpha[c])
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            if not (abs(ll[c] - lr) <= RTOL * max(sc, abs(lr)) or (np.isnan(lr) and np.isnan(ll[c])) or (np.isinf(lr) and lr == ll[c])):
                bad += 1
                if bad < 4: print('  loglike cell', c, cnt[c], ll[c], lr)
            try:
                br = cl.get_beta_max(alpha[c], guess=guess)
            except ValueError:
                nerr += 1
                if not (fl[c] & 0x10):
                    bad += 1
                    if bad < 4: print('  cell', c, 'oracle raised; flags', fl[c], beta[c])
                continue
            e = abs(beta[c] - br) / max(1, abs(br))
            worst = max(worst, e)
            if not e <= RTOL:
                bad += 1
                if bad < 4: print('  beta cell', c, 'n', cnt[c], 'alpha', alpha[c], beta[c], br, 'flags', fl[c])
                continue
            lpr = cl.log_profile_likelihood(alpha[c], beta_guess=guess)
            if not (abs(lp[c] - lpr) <= RTOL * max(sc, abs(lpr)) or (np.isnan(lpr) and np.isnan(lp[c])) or lpr == lp[c]):
                bad += 1
                if bad < 4: print('  lprof cell', c, 'n', cnt[c], lp[c], lpr)
    nbad += bad
    print('conf %d mu %g %s %s guess %g: bad %d worst %.2e oracle errors %d [%.0f s]' % (conf, mu, cdist, wd_, guess, bad, worst, nerr, time.time() - t0), flush=True)
print('TOTAL BAD', nbad)
