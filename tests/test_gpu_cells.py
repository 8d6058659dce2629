"""GPU parity of the per-cell likelihood kernels against the reference's
golden vectors and against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import godot_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-9     # the tolerance the path is specified to


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


def _ranges(g, dev):
    from godot_amd import ops
    off = np.append(0, np.cumsum(g['counts'])).astype(np.int64)
    S = g['exp']
    B = g['exp'] / g['SonB']
    cr = ops.CellRanges.from_offsets(
        torch.from_numpy(g['we']).to(dev), torch.from_numpy(off).to(dev),
        torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    return cr, off


def _close(a, b, scale=None, rtol=RTOL):
    a = np.asarray(a)
    b = np.asarray(b)
    tol = rtol * np.maximum(np.abs(b), 1.0 if scale is None else scale)
    bad = ~((np.abs(a - b) <= tol) | (np.isnan(a) & np.isnan(b)))
    assert not bad.any(), 'mismatch at %s: got %s want %s' % (
        np.argwhere(bad)[:5].ravel(), a[bad][:5], b[bad][:5])


def _logscale(g, off):
    """sum |log(1-w)| per cell: the size of the terms TS / logL are sums of."""
    s = np.add.reduceat(np.append(np.abs(np.log1p(-g['we'])), 0.), off[:-1])
    s[np.diff(off) == 0] = 0
    return np.maximum(s, 1.0)


@pytest.mark.parametrize('name', ['cells_dense.npz', 'cells_sparse.npz', 'cells_edge.npz'])
def test_fit_and_ts_match_reference(golden, name):
    from godot_amd import ops
    g = golden(name)
    cr, off = _ranges(g, _dev())
    alpha, ts, flags = ops.cell_fit(cr)
    alpha, ts, flags = alpha.cpu().numpy(), ts.cpu().numpy(), flags.cpu().numpy()
    _close(alpha, g['alpha'])
    _close(ts, g['ts'], scale=_logscale(g, off))
    # cells where the reference raised are flagged, not silently wrong
    assert np.array_equal((flags & 0x10) != 0, np.isnan(g['alpha']))
    assert np.array_equal((flags & 0x01) != 0, g['alpha'] == 0)


def test_fit_from_guesses(golden):
    from godot_amd import ops
    g = golden('cells_edge.npz')
    cr, off = _ranges(g, _dev())
    for j, gs in enumerate(g['guesses']):
        a, _, _ = ops.cell_fit(cr, guess=float(gs), want_ts=False)
        _close(a.cpu().numpy(), g['alpha_from_guess'][:, j])
        gt = torch.full((cr.ncell,), float(gs), dtype=torch.float64, device=cr.device)
        a2, _, _ = ops.cell_fit(cr, guess=gt, want_ts=False)
        assert torch.equal(a, a2) or np.array_equal(a.cpu().numpy(), a2.cpu().numpy(), equal_nan=True)


@pytest.mark.parametrize('name', ['cells_dense.npz', 'cells_sparse.npz', 'cells_edge.npz'])
def test_loglike_matches_reference(golden, name):
    from godot_amd import ops
    g = golden(name)
    cr, off = _ranges(g, _dev())
    al = torch.full((cr.ncell,), 1.3, dtype=torch.float64, device=cr.device)
    ll = ops.cell_loglike(cr, al).cpu().numpy()
    _close(ll, g['logl_1p3'], scale=_logscale(g, off))


@pytest.mark.parametrize('name', ['cells_dense.npz', 'cells_sparse.npz', 'cells_edge.npz'])
def test_logpdf_matches_reference(golden, name):
    from godot_amd import ops
    g = golden(name)
    cr, off = _ranges(g, _dev())
    dom, cod, flags = ops.cell_logpdf(cr, dlogl=30, npt=200)
    n = len(g['dom'])
    dom, cod = dom.cpu().numpy()[:n], cod.cpu().numpy()[:n]
    ok = ~np.isnan(g['alpha'][:n])
    scale = _logscale(g, off)[:n, None]
    _close(dom[ok], g['dom'][ok])
    _close(cod[ok], g['cod'][ok], scale=np.broadcast_to(scale, cod.shape)[ok])
    assert np.isnan(dom[~ok]).all()


@pytest.mark.parametrize('name,n', [('cells_dense.npz', None), ('cells_sparse.npz', 300)])
def test_lightcurve_matches_reference(golden, name, n):
    from godot_amd import ops
    g = golden(name)
    cr, off = _ranges(g, _dev())
    n = n or cr.ncell
    dev = cr.device
    dom, cod, _ = ops.cell_logpdf(cr, dlogl=30, npt=200)
    ag, xlo, xhi = ops.profile_quantiles(dom, cod, 0.159, 0.841)
    alpha, ts, _ = ops.cell_fit(cr, guess=ag)
    rows = ops.lightcurve_rows(torch.from_numpy(g['starts']).to(dev),
                               torch.from_numpy(g['stops']).to(dev), alpha, ts, xlo, xhi, cr.S, 4.)
    rows = rows.cpu().numpy()[:n]
    ref = g['lightcurve']
    sc = _logscale(g, off)[:n]
    _close(rows[:, 0], ref[:, 1])
    # upper-limit / detection branch must agree exactly
    assert np.array_equal(rows[:, 3] == -1, ref[:, 4] == -1)
    _close(rows[:, 1], ref[:, 2])
    _close(rows[:, 2], ref[:, 3])
    _close(rows[:, 3], ref[:, 4])
    _close(rows[:, 4], ref[:, 5], scale=sc)


def test_beta_max_and_profile_match_reference(golden):
    from godot_amd import ops
    g = golden('profile_bkg.npz')
    cr, off = _ranges(g, _dev())
    sc = _logscale(g, off)
    for j, al in enumerate(g['alphas']):
        beta, lp, flags = ops.cell_beta_max(cr, float(al))
        beta, lp, flags = beta.cpu().numpy(), lp.cpu().numpy(), flags.cpu().numpy()
        ref = g['beta_max'][:, j]
        _close(beta, ref)
        _close(lp, g['logl_profile'][:, j], scale=sc)
        assert np.array_equal((flags & 0x10) != 0, np.isnan(ref))


def test_moments_match_oracle(golden):
    from godot_amd import ops
    g = golden('cells_dense.npz')
    cr, off = _ranges(g, _dev())
    w1, w2 = ops.cell_moments(cr.we, cr.lo, cr.hi)
    r1 = np.add.reduceat(np.append(g['we'], 0.), off[:-1])
    r2 = np.add.reduceat(np.append(g['we'] ** 2, 0.), off[:-1])
    _close(w1.cpu().numpy(), r1, rtol=1e-12)
    _close(w2.cpu().numpy(), r2, rtol=1e-12)


def test_bad_ranges_are_flagged_not_read():
    from godot_amd import ops
    dev = _dev()
    we = torch.rand(100, dtype=torch.float64, device=dev) * 0.9 + 0.05
    lo = torch.tensor([0, 50, 90, -5], dtype=torch.int64, device=dev)
    hi = torch.tensor([50, 40, 200, 10], dtype=torch.int64, device=dev)
    S = torch.full((4,), 5., dtype=torch.float64, device=dev)
    cr = ops.CellRanges(we, lo, hi, S, S.clone())
    a, ts, fl = ops.cell_fit(cr)
    a, fl = a.cpu().numpy(), fl.cpu().numpy()
    assert np.isfinite(a[0]) and np.isnan(a[1:]).all()
    assert (fl[1:] & 0x10).all()
    # a fault of the caller's, told apart from a cell the data make unfittable, and never
    # passed over: not even where unfittable cells are allowed
    assert (fl[1:] & 0x80).all() and not fl[0] & 0x80
    from godot_amd import core
    from godot_amd._lib import GodotError
    for allow in (False, True):
        with pytest.raises(GodotError):
            core._raise_as_reference(torch.from_numpy(fl).to(dev), allow)
    core._raise_as_reference(torch.from_numpy(fl[:1]).to(dev), False)
    for shape in ((3000,), (40,)):   # one workgroup per cell; lane groups
        n = shape[0]
        w2 = torch.rand(2 * n, dtype=torch.float64, device=dev) * 0.9 + 0.05
        lo2 = torch.tensor([0, n], dtype=torch.int64, device=dev)
        hi2 = torch.tensor([n, 2 * n + 1], dtype=torch.int64, device=dev)
        S2 = torch.full((2,), n * 0.5, dtype=torch.float64, device=dev)
        _, _, f2 = ops.cell_fit(ops.CellRanges(w2, lo2, hi2, S2, S2.clone()))
        f2 = f2.cpu().numpy()
        assert f2[1] & 0x80 and f2[1] & 0x10 and not f2[0] & 0x90
        _, _, f3 = ops.cell_logpdf(ops.CellRanges(w2, lo2, hi2, S2, S2.clone()), npt=50)
        f3 = f3.cpu().numpy()
        assert f3[1] & 0x80 and not f3[0] & 0x80


@pytest.mark.parametrize('nph,ncell', [(200000, 20000), (200000, 700), (100000, 3)])
def test_fit_matches_oracle_on_random_cells(nph, ncell):
    """Ragged random cells (incl. empty ones) across the three team sizes."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(nph + ncell)
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    cuts = np.sort(rng.integers(0, nph + 1, ncell - 1))
    off = np.concatenate([[0], cuts, [nph]]).astype(np.int64)
    cnt = np.diff(off)
    S = np.maximum(cnt * 0.17 * rng.uniform(0.3, 3, ncell), 0.05)
    B = S * 5
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    alpha, ts, flags = ops.cell_fit(cr)
    alpha, ts = alpha.cpu().numpy(), ts.cpu().numpy()
    pick = rng.choice(ncell, min(ncell, 400), replace=False)
    with np.errstate(all='ignore'):
        for c in pick:
            cl = orc.CellLike(we[off[c]:off[c + 1]], S[c], S[c] / B[c])
            a = cl.get_max()
            t = cl.get_ts(a)
            sc = max(1.0, np.abs(np.log1p(-we[off[c]:off[c + 1]])).sum())
            assert abs(alpha[c] - a) <= RTOL * max(1, abs(a)), (c, alpha[c], a)
            assert abs(ts[c] - t) <= RTOL * max(sc, abs(t)), (c, ts[c], t)


@pytest.mark.parametrize('mean,variant', [
    (4, 'plain'), (4, 'guess'), (9, 'plain'), (12.5, 'plain'), (12.5, 'beta'), (12.5, 'scattered'), (17, 'plain'),
    # one lane of 24 slots, restarts by the second kernel; two lanes of 20 and of 32 slots
    (11, 'plain'), (11, 'guess'), (11, 'swap'), (25, 'plain'), (25, 'beta'), (25, 'outliers'), (36, 'plain'), (36, 'swap'),
    (30, 'plain'), (30, 'beta'), (30, 'guess'),
    (22, 'plain'), (22, 'guess'), (22, 'beta'), (22, 'scattered'), (22, 'swap'), (40, 'plain'), (40, 'beta'),
    (22, 'outliers'),
    # four lanes of 20 .. 32 slots, eight of 20 and 24, sixteen of 16 .. 24, thirty-two and sixty-four of 16 .. 24
    (62, 'plain'), (62, 'guess'), (75, 'plain'), (75, 'swap'), (90, 'plain'), (90, 'beta'), (105, 'plain'),
    (105, 'scattered'), (135, 'plain'), (135, 'outliers'), (165, 'plain'), (165, 'guess'), (220, 'plain'),
    (220, 'beta'), (280, 'plain'), (280, 'swap'), (340, 'plain'), (450, 'plain'), (450, 'guess'), (580, 'plain'),
    (700, 'plain'), (700, 'beta'), (900, 'plain'), (900, 'scattered'), (1150, 'plain'), (1400, 'plain'),
    (1400, 'guess'),
    # the largest shape: cells inside its 2 048 slots (the fit leaves those to the team kernel) and beyond them
    (1700, 'plain'), (2300, 'plain'), (2300, 'beta'), (2700, 'plain'), (2700, 'guess'), (2700, 'swap')])
def test_fit_sparse_cell_kernels(mean, variant):
    """Every shape of the lane-group kernel (1 to 64 lanes per cell; restarts
    in place or by the second kernel), with per-cell guesses, a background
    level other than 1, cells that are not contiguous in memory, complemented
    weights, and a few cells far larger than the rest."""
    from godot_amd import ops
    dev = _dev()
    ncell = 6000 if mean <= 40 else max(700, int(6000 * 40 / mean))
    rng = np.random.default_rng(int(mean * 10) + len(variant))
    cnt = rng.poisson(mean, ncell).astype(np.int64)
    if variant == 'outliers':
        cnt[rng.choice(ncell, 12, replace=False)] = rng.integers(100, 3000, 12)
        cnt[:3] = 0
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    nph = int(off[-1])
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    S = np.maximum(cnt * 0.17 * rng.uniform(0.3, 3, ncell), 0.05)
    B = S * 5
    lo, hi = off[:-1].copy(), off[1:].copy()
    order = np.arange(ncell)
    if variant == 'scattered':
        order = rng.permutation(ncell)
    kw = {}
    guess = np.ones(ncell)
    beta = 1.0
    if variant == 'guess':
        guess = rng.uniform(0.2, 4.0, ncell)
        kw['guess'] = torch.from_numpy(guess).to(dev)
    if variant == 'beta':
        beta = 1.7
        kw['beta'] = beta
    wd = torch.from_numpy(we).to(dev)
    Sd, Bd = torch.from_numpy(S[order]).to(dev), torch.from_numpy(B[order]).to(dev)
    if variant == 'swap':
        # complemented weights with S and B exchanged describe the same cells
        cr = ops.CellRanges(1 - wd, torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev), Bd, Sd)
        kw['swap'] = True
    else:
        cr = ops.CellRanges(wd, torch.from_numpy(lo[order]).to(dev), torch.from_numpy(hi[order]).to(dev), Sd, Bd)
    alpha, ts, flags = ops.cell_fit(cr, **kw)
    alpha, ts, flags = alpha.cpu().numpy(), ts.cpu().numpy(), flags.cpu().numpy()
    assert not (flags & 0x10).any()
    depth = (flags >> 1) & 3
    big = np.flatnonzero(cnt[order] > 64) if mean <= 40 else np.flatnonzero(cnt[order] > mean + 5 * np.sqrt(mean))
    pick = np.unique(np.concatenate([rng.choice(ncell, 300 if mean <= 40 else 120, replace=False),
                                     np.flatnonzero(depth > 0)[:150], big[:40]]))
    with np.errstate(all='ignore'):
        for i in pick:
            c = order[i]
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            a = cl.get_max(guess=guess[i], beta=beta)
            t = cl.get_ts(a)
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            tol = RTOL * (100 if variant == 'swap' else 1)   # 1 - (1 - w) is not w to the last bit
            assert abs(alpha[i] - a) <= tol * max(1, abs(a)), (i, cnt[c], depth[i], alpha[i], a)
            assert abs(ts[i] - t) <= tol * max(sc, abs(t)), (i, cnt[c], depth[i], ts[i], t)
            assert bool(flags[i] & 1) == (a == 0)
    if variant == 'plain' and mean <= 40:
        assert (depth > 0).any()        # the restart path was exercised


@pytest.mark.parametrize('mean,variant', [(4, 'plain'), (9, 'plain'), (12.5, 'plain'), (17, 'plain'), (30, 'plain'),
                                          (22, 'plain'), (22, 'swap'),
                                          (22, 'outliers'), (22, 'given'), (40, 'plain'), (22, 'npt300'),
                                          (62, 'plain'), (75, 'plain'), (90, 'swap'), (105, 'plain'),
                                          (135, 'given'), (165, 'plain'), (220, 'plain'), (280, 'outliers'),
                                          (340, 'plain'), (450, 'plain'), (580, 'given'), (700, 'plain'),
                                          (900, 'plain'), (1150, 'swap'), (1400, 'plain'), (1400, 'npt300'),
                                          (1700, 'plain'), (2300, 'plain'), (2700, 'plain'), (2700, 'given'),
                                          (2700, 'swap')])
def test_logpdf_sparse_cell_kernels(mean, variant):
    """Profile grids of the sparse-cell kernel in every shape against the
    oracle's get_logpdf: centre found by the kernel or given, complemented
    weights, a few cells far larger than the rest, more points than one
    sweep of the wavefront holds."""
    from godot_amd import ops
    dev = _dev()
    ncell = 3000 if mean <= 40 else max(400, int(3000 * 40 / mean))
    rng = np.random.default_rng(int(mean * 10) + len(variant) + 77)
    cnt = rng.poisson(mean, ncell).astype(np.int64)
    if variant == 'outliers':
        cnt[rng.choice(ncell, 8, replace=False)] = rng.integers(100, 2500, 8) * (1 if mean <= 40 else 3)
        cnt[:3] = 0
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
    S = np.maximum(cnt * 0.17 * rng.uniform(0.3, 3, ncell), 0.05)
    B = S * 5
    npt = 300 if variant == 'npt300' else 200
    wd = torch.from_numpy(we).to(dev)
    od = torch.from_numpy(off).to(dev)
    Sd, Bd = torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev)
    kw = {}
    if variant == 'swap':
        cr = ops.CellRanges.from_offsets(1 - wd, od, Bd, Sd)
        kw['swap'] = True
    else:
        cr = ops.CellRanges.from_offsets(wd, od, Sd, Bd)
    given = None
    if variant == 'given':
        if mean <= 40:
            given = rng.uniform(0.0, 3.0, ncell)
            given[::7] = 0.0
        else:
            # the profile of a large cell is narrow: a centre far from its maximum is re-centred
            # without end by the reference (a RecursionError there, a failed cell here)
            given = ops.cell_fit(cr)[0].cpu().numpy() * rng.uniform(0.97, 1.03, ncell)
        kw['aopt'] = torch.from_numpy(given).to(dev)
    dom, cod, flags = ops.cell_logpdf(cr, dlogl=30, npt=npt, **kw)
    dom, cod, flags = dom.cpu().numpy(), cod.cpu().numpy(), flags.cpu().numpy()
    assert not (flags & 0x10).any()
    big = np.flatnonzero(cnt > 64) if mean <= 40 else np.flatnonzero(cnt > mean + 5 * np.sqrt(mean))
    pick = np.unique(np.concatenate([rng.choice(ncell, 120 if mean <= 40 else 70, replace=False),
                                     np.flatnonzero(flags & 0x20)[:40], big[:20], np.arange(3)]))
    tol = RTOL * (100 if variant == 'swap' else 1)
    ncmp = 0
    with np.errstate(all='ignore'):
        for c in pick:
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            d, v = cl.get_logpdf(aopt=None if given is None else given[c], dlogl=30, npt=npt)
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            if np.isnan(d).any() or np.isnan(v).any():
                continue      # the reference's own Newton steps left the domain: nothing to compare
            assert np.abs(dom[c] - d).max() <= tol * max(1.0, np.abs(d).max()), (c, cnt[c], flags[c])
            assert np.abs(cod[c] - v).max() <= tol * max(sc, np.abs(v).max()), (c, cnt[c], flags[c])
            ncmp += 1
    assert ncmp > 60


def test_fits_on_different_streams_give_the_same_bits():
    """Independent calls may overlap on different streams (the entry points keep no state
    between calls): fits and profile grids of three lists launched on three streams, several
    times over, equal the same calls made one after the other to the last bit."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(17)
    crs = []
    for mean, ncell in ((22, 40000), (9, 60000), (120, 9000)):
        cnt = rng.poisson(mean, ncell).astype(np.int64)
        off = np.append(0, np.cumsum(cnt)).astype(np.int64)
        we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
        S = np.maximum(cnt * 0.17 * rng.uniform(0.3, 3, ncell), 0.05)
        crs.append(ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                               torch.from_numpy(S).to(dev), torch.from_numpy(S * 5).to(dev)))
    ref = [ops.cell_fit(cr) + ops.cell_logpdf(cr, dlogl=30, npt=50) for cr in crs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev) for _ in crs]
    for rep in range(4):
        outs = []
        for cr, st in zip(crs, streams):
            with torch.cuda.stream(st):
                outs.append(ops.cell_fit(cr) + ops.cell_logpdf(cr, dlogl=30, npt=50))
        torch.cuda.synchronize()
        for got, want in zip(outs, ref):
            for g, w in zip(got, want):
                assert torch.equal(g.view(torch.int64) if g.dtype == torch.float64 else g,
                                   w.view(torch.int64) if w.dtype == torch.float64 else w)


@pytest.mark.parametrize('swap', [False, True])
def test_logpdf_of_the_last_cells_of_the_photon_array(swap):
    """The grid evaluation asks for the last n mod 8 weights of a cell in one request of eight,
    which reads on into the next cell; the cells at the very end of the array have nothing
    behind them and take their weights one by one.  Every remainder 0 .. 7 at the end of the
    array, every cell against the oracle."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(5)
    for tail in range(0, 20):
        cnt = np.concatenate([rng.poisson(13, 40), [tail]]).astype(np.int64)
        off = np.append(0, np.cumsum(cnt)).astype(np.int64)
        we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
        S = np.maximum(cnt * 0.17 * rng.uniform(0.3, 3, len(cnt)), 0.05)
        B = S * 5
        wd, od = torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev)
        Sd, Bd = torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev)
        if swap:
            cr = ops.CellRanges.from_offsets(1 - wd, od, Bd, Sd)
        else:
            cr = ops.CellRanges.from_offsets(wd, od, Sd, Bd)
        dom, cod, flags = ops.cell_logpdf(cr, dlogl=30, npt=200, swap=swap)
        dom, cod, flags = dom.cpu().numpy(), cod.cpu().numpy(), flags.cpu().numpy()
        assert not (flags & 0x10).any()
        tol = RTOL * (100 if swap else 1)
        with np.errstate(all='ignore'):
            for c in range(len(cnt) - 6, len(cnt)):
                w = we[off[c]:off[c + 1]]
                d, v = orc.CellLike(w, S[c], S[c] / B[c]).get_logpdf(dlogl=30, npt=200)
                if np.isnan(d).any() or np.isnan(v).any():
                    continue
                sc = max(1.0, np.abs(np.log1p(-w)).sum())
                assert np.abs(dom[c] - d).max() <= tol * max(1.0, np.abs(d).max()), (tail, c, cnt[c])
                assert np.abs(cod[c] - v).max() <= tol * max(sc, np.abs(v).max()), (tail, c, cnt[c])


@pytest.mark.parametrize('variant', ['plain', 'swap', 'beta', 'guess', 'npt300', 'given'])
def test_very_large_cells(variant):
    """Cells of tens of thousands of photons (what merged Bayesian blocks are):
    one workgroup per cell; fit, TS and profile grid against the oracle."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(len(variant) + 31)
    cnt = np.array([20000, 70001, 33000, 17003], dtype=np.int64)
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
    S = cnt * 0.17 * np.array([1.0, 0.4, 2.5, 40.0])
    B = S * 5
    wd, od = torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev)
    Sd, Bd = torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev)
    kw, gkw = {}, {}
    guess, beta = 1.0, 1.0
    if variant == 'swap':
        cr = ops.CellRanges.from_offsets(1 - wd, od, Bd, Sd)
        kw['swap'] = gkw['swap'] = True
    else:
        cr = ops.CellRanges.from_offsets(wd, od, Sd, Bd)
    if variant == 'beta':
        beta = kw['beta'] = 1.6
    if variant == 'guess':
        guess = kw['guess'] = 0.3
    npt = 300 if variant == 'npt300' else 100
    given = None
    if variant == 'given':
        given = np.array([0.9, 2.3, 0.0, 0.05])
        gkw['aopt'] = torch.from_numpy(given).to(dev)
    alpha, ts, flags = ops.cell_fit(cr, **kw)
    dom, cod, gfl = ops.cell_logpdf(cr, dlogl=20, npt=npt, **gkw)
    alpha, ts, dom, cod = alpha.cpu().numpy(), ts.cpu().numpy(), dom.cpu().numpy(), cod.cpu().numpy()
    tol = RTOL * (100 if variant == 'swap' else 1)
    with np.errstate(all='ignore'):
        for c in range(len(cnt)):
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            a = cl.get_max(guess=guess, beta=beta)
            t = cl.get_ts(a)
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            assert abs(alpha[c] - a) <= tol * max(1, abs(a)), (c, alpha[c], a)
            assert abs(ts[c] - t) <= tol * max(sc, abs(t)), (c, ts[c], t)
            try:
                d, v = cl.get_logpdf(aopt=None if given is None else given[c], dlogl=20, npt=npt)
            except RuntimeError:
                # a centre far from the maximum: the reference re-centres without end
                assert gfl[c].item() & 0x10 and np.isnan(dom[c]).all()
                continue
            if np.isnan(d).any() or np.isnan(v).any():
                continue
            assert np.abs(dom[c] - d).max() <= tol * max(1.0, np.abs(d).max()), c
            assert np.abs(cod[c] - v).max() <= tol * max(sc, np.abs(v).max()), c


@pytest.mark.parametrize('variant', ['swap', 'beta', 'guess'])
def test_streamed_cells_in_the_variants_of_the_fit(variant):
    """Cells of 3 100 to 6 400 photons, which one wavefront streams in every pass of the fit:
    the complemented weights, a background level other than 1 and a guess other than 1."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(len(variant) + 57)
    cnt = rng.integers(3100, 6400, 300).astype(np.int64)
    cnt[:4] = [3100, 3101, 3102, 3103]
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
    S = cnt * 0.17 * rng.choice([0.4, 1.0, 2.5], len(cnt))
    B = S * 5
    wd, od = torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev)
    Sd, Bd = torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev)
    kw = {}
    guess, beta = 1.0, 1.0
    if variant == 'swap':
        cr = ops.CellRanges.from_offsets(1 - wd, od, Bd, Sd)
        kw['swap'] = True
    else:
        cr = ops.CellRanges.from_offsets(wd, od, Sd, Bd)
    if variant == 'beta':
        beta = kw['beta'] = 1.6
    if variant == 'guess':
        guess = kw['guess'] = 0.3
    alpha, ts, flags = ops.cell_fit(cr, **kw)
    alpha, ts = alpha.cpu().numpy(), ts.cpu().numpy()
    assert not (flags.cpu().numpy() & 0x10).any()
    tol = RTOL * (100 if variant == 'swap' else 1)
    with np.errstate(all='ignore'):
        for c in [0, 1, 2, 3, 77, 150, 299]:
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            a = cl.get_max(guess=guess, beta=beta)
            t = cl.get_ts(a)
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            assert abs(alpha[c] - a) <= tol * max(1, abs(a)), (c, alpha[c], a)
            assert abs(ts[c] - t) <= tol * max(sc, abs(t)), (c, ts[c], t)


def test_zero_alpha_is_flagged_however_it_was_reached():
    """alpha-hat = 0 comes about in two ways: the derivative at zero is negative, or a
    Newton/Halley iterate is clamped to the boundary (core.py:671, 680).  Both carry the
    at-zero flag, and both agree with the oracle.  Heavy-tailed cell sizes with a strong
    source term produce hundreds of each."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(12)
    ncell = 10000
    cnt = (rng.geometric(1.0 / 61, ncell) - 1).astype(np.int64)
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    n = int(off[-1])
    we = np.clip(np.where(rng.random(n) < 0.5, rng.beta(0.3, 1.5, n), rng.beta(5, 0.5, n)), 1e-7, 1 - 1e-7)
    S = 60 * we.mean() * 3 * rng.uniform(0.7, 1.3, ncell)
    B = 60 * (1 - we.mean()) * rng.uniform(0.7, 1.3, ncell)
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    alpha, ts, flags = ops.cell_fit(cr)
    alpha, ts, flags = alpha.cpu().numpy(), ts.cpu().numpy(), flags.cpu().numpy()
    assert np.array_equal((flags & 0x01) != 0, alpha == 0)
    assert not (flags & 0x10).any()
    clamped = 0
    with np.errstate(all='ignore'):
        for c in range(ncell):
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            a = cl.get_max()
            assert abs(alpha[c] - a) <= RTOL * max(1, abs(a)), (c, alpha[c], a)
            assert (a == 0) == (alpha[c] == 0), c
            if a == 0 and len(w) and (w / (1 - w)).sum() - S[c] >= 0:
                clamped += 1        # the derivative at zero did not send it there
    assert (alpha == 0).sum() > 300 and clamped > 10


@pytest.mark.parametrize('shape', ['spread64', 'one_workgroup', 'moderate_few', 'streamed'])
def test_very_large_cells_by_layout(shape):
    """The layouts of very large cells: few cells, each spread over many workgroups
    that exchange partial sums (3 cells of ~600 000 photons); more cells than half the
    compute units, one workgroup each (150 cells of ~17 000); and a few hundred cells of
    5 000 to 9 000 photons (three-day cells of a bright source), which also get a
    workgroup each because one wavefront per cell would stream them too slowly; and cells of
    3 100 to 6 400, beyond the registers of a wavefront, whose fit one wavefront streams
    four loads at a time."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(5 + len(shape))
    if shape == 'spread64':
        cnt = np.array([600001, 580000, 655003], dtype=np.int64)
        check = [0, 1, 2]
    elif shape == 'one_workgroup':
        cnt = rng.integers(16500, 18000, 150).astype(np.int64)
        check = [0, 7, 77, 149]
    elif shape == 'moderate_few':
        cnt = rng.integers(5200, 9000, 300).astype(np.int64)
        check = [0, 13, 150, 299]
    else:
        cnt = rng.integers(3100, 6400, 300).astype(np.int64)
        cnt[:4] = [3100, 3101, 3102, 3103]      # ragged ends of the four-load batches
        check = [0, 1, 2, 3, 150, 299]
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
    S = cnt * 0.17 * rng.choice([0.4, 1.0, 2.5], len(cnt))
    B = S * 5
    wd, od = torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev)
    cr = ops.CellRanges.from_offsets(wd, od, torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    alpha, ts, flags = ops.cell_fit(cr)
    dom, cod, gfl = ops.cell_logpdf(cr, dlogl=20, npt=100)
    # a second call reuses the exchange slots: same bits
    alpha2, ts2, _ = ops.cell_fit(cr)
    dom2, cod2, _ = ops.cell_logpdf(cr, dlogl=20, npt=100)
    assert torch.equal(alpha, alpha2) and torch.equal(ts, ts2)
    assert torch.equal(dom, dom2) and torch.equal(cod, cod2)
    alpha, ts, dom, cod = alpha.cpu().numpy(), ts.cpu().numpy(), dom.cpu().numpy(), cod.cpu().numpy()
    assert not (flags.cpu().numpy() & 0x10).any() and not (gfl.cpu().numpy() & 0x10).any()
    with np.errstate(all='ignore'):
        for c in check:
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            a = cl.get_max()
            t = cl.get_ts(a)
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            assert abs(alpha[c] - a) <= RTOL * max(1, abs(a)), (c, alpha[c], a)
            assert abs(ts[c] - t) <= RTOL * max(sc, abs(t)), (c, ts[c], t)
            d, v = cl.get_logpdf(dlogl=20, npt=100)
            assert np.abs(dom[c] - d).max() <= RTOL * max(1.0, np.abs(d).max()), c
            assert np.abs(cod[c] - v).max() <= RTOL * max(sc, np.abs(v).max()), c


def test_spread_cells_whose_workgroups_give_up_waiting_are_fitted_again():
    """The workgroups of a spread cell wait for each other a bounded time (they are resident
    together unless the GPU is shared or busy).  With the wait cut to one poll nearly every
    exchange runs out: the cells are marked, fitted again by one workgroup each, and come out
    with the results of an undisturbed run and the GODOT_F_RETRIED bit, never failed."""
    from godot_amd import ops, _lib
    dev = _dev()
    rng = np.random.default_rng(17)
    cnt = np.array([300001, 280000, 355003, 290000], dtype=np.int64)
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
    S = cnt * 0.17 * np.array([0.4, 1.0, 2.5, 1.0])
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(S * 5).to(dev))
    a0, t0, f0 = ops.cell_fit(cr)
    d0, c0, g0 = ops.cell_logpdf(cr, dlogl=20, npt=100)
    torch.cuda.synchronize()
    assert not ((f0 | g0).cpu().numpy() & 0x50).any()          # neither failed nor retried
    try:
        _lib.call('godot_debug_spread_spin_limit', 1)
        a1, t1, f1 = ops.cell_fit(cr)
        d1, c1, g1 = ops.cell_logpdf(cr, dlogl=20, npt=100)
        torch.cuda.synchronize()
    finally:
        _lib.call('godot_debug_spread_spin_limit', 0)
    f1, g1 = f1.cpu().numpy(), g1.cpu().numpy()
    assert not ((f1 | g1) & 0x10).any()
    assert (f1 & 0x40).any() and (g1 & 0x40).any()             # the retry path did run
    rel = lambda x, y: float(((x - y).abs() / y.abs().clamp(min=1.0)).max())
    assert rel(a1, a0) < 1e-12 and rel(t1, t0) < 1e-10
    assert rel(d1, d0) < 1e-12 and rel(c1, c0) < 1e-10
    # and the default wait is back: an undisturbed run again
    a2, t2, f2 = ops.cell_fit(cr)
    assert torch.equal(a2, a0) and torch.equal(t2, t0) and not (f2.cpu().numpy() & 0x40).any()


@pytest.mark.parametrize('given', [False, True])
def test_profiles_of_few_moderately_large_cells(given):
    """Up to 4096 cells of 1024 photons or more each (ten years of daily cells of a bright
    source): the profile grid takes one workgroup per cell, like very large cells, because
    the dense-cell kernel would leave most of the GPU idle.  Ragged sizes, among them an
    empty cell and cells far below the mean; with and without given centres."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(31 + given)
    cnt = rng.integers(1100, 4000, 400).astype(np.int64)
    cnt[[3, 50, 200]] = [0, 7, 90]
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    assert off[-1] >= 1024 * len(cnt)
    we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
    S = np.maximum(cnt, 1) * 0.17 * rng.choice([0.05, 0.4, 1.0, 2.5], len(cnt))
    B = S * 5
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    aopt = None
    if given:
        aopt = ops.cell_fit(cr, want_ts=False)[0] * torch.from_numpy(rng.choice([1.0, 0.8, 1.3], len(cnt))).to(dev)
    dom, cod, fl = ops.cell_logpdf(cr, aopt=aopt, dlogl=30, npt=200)
    dom, cod, fl = dom.cpu().numpy(), cod.cpu().numpy(), fl.cpu().numpy()
    a_h = None if aopt is None else aopt.cpu().numpy()
    failed = np.flatnonzero(fl & 0x10)
    if not given:
        assert len(failed) == 0
    with np.errstate(all='ignore'):
        for c in [0, 3, 50, 200] + list(rng.choice(len(cnt), 20, replace=False)) + list(failed[:3]):
            w = we[off[c]:off[c + 1]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            try:
                d, v = cl.get_logpdf(aopt=None if a_h is None else a_h[c], dlogl=30, npt=200)
            except RuntimeError:
                # far from the maximum of a cell with alpha-hat ~ 77 the reference re-centres
                # without end (core.py:923-929): flagged here, NaN rows
                assert fl[c] & 0x10 and np.isnan(dom[c]).all() and np.isnan(cod[c]).all(), c
                continue
            assert not (fl[c] & 0x10), c
            if np.isnan(d).all():
                # the search for an end of the support left the domain of the logarithm:
                # NaN throughout in the reference, without an exception
                assert np.isnan(dom[c]).all() and np.isnan(cod[c]).all(), c
                continue
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            assert np.abs(dom[c] - d).max() <= RTOL * max(1.0, np.abs(d).max()), c
            assert np.abs(cod[c] - v).max() <= RTOL * max(sc, np.abs(v).max()), c


def test_cells_the_reference_cannot_fit_raise_the_same_error(golden):
    """Cells at the edge of an exposure gap (hundreds of photons against an expected 0.1):
    alpha-hat lies beyond the bracket of the reference's root finder and get_max raises
    ValueError (core.py:728-739).  The kernels flag such a cell and give NaN; the batch
    class raises what the reference raises, or goes on when asked to.  Fixture: cells the
    oracle found in a randomised light curve (tools/make_golden_unfittable.py)."""
    from godot_amd import core, ops
    dev = _dev()
    g = golden('unfittable_cells.npz')
    ws = [g['we%d' % k] for k in range(int(g['n']))]
    # two ordinary cells around them
    rng = np.random.default_rng(0)
    ws = [np.clip(rng.beta(0.3, 1.5, 40), 1e-6, 1 - 1e-6)] + ws + [np.clip(rng.beta(0.3, 1.5, 25), 1e-6, 1 - 1e-6)]
    S = np.array([7.0] + [float(g['S%d' % k]) for k in range(int(g['n']))] + [4.0])
    B = np.array([33.0] + [float(g['B%d' % k]) for k in range(int(g['n']))] + [21.0])
    off = np.append(0, np.cumsum([len(w) for w in ws])).astype(np.int64)
    we = np.concatenate(ws)
    for k in range(1, len(ws) - 1):
        with pytest.raises(ValueError), np.errstate(all='ignore'):
            orc.CellLike(ws[k], S[k], S[k] / B[k]).get_max()
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    alpha, ts, flags = ops.cell_fit(cr)
    alpha, flags = alpha.cpu().numpy(), flags.cpu().numpy()
    assert np.isnan(alpha[1:-1]).all() and np.isfinite(alpha[[0, -1]]).all()
    assert ((flags[1:-1] & 0x18) == 0x18).all() and not (flags[[0, -1]] & 0x10).any()
    starts = np.arange(len(ws)) * 100.0
    batch = core.CellBatch(cr.we, cr.we, cr.lo, cr.hi, torch.from_numpy(starts).to(dev),
                           torch.from_numpy(starts + 100.0).to(dev), cr.S, cr.B)
    with pytest.raises(ValueError, match='different signs'):
        core.CellsLogLikelihood(batch)
    cll = core.CellsLogLikelihood(batch, allow_failed_cells=True)
    lc = cll.get_raw_lightcurve()
    assert np.isnan(lc[1:-1, 2:]).all() and np.isfinite(lc[[0, -1]]).all()


def test_profile_of_only_nan_takes_its_place_in_the_light_curve(golden):
    """A cell with alpha-hat = 0 whose search for the upper end of the support leaves the
    domain of the logarithm (core.py:869-880): the reference does not stop, its profile is
    NaN throughout, and get_raw_lightcurve (core.py:1295-1336) writes the row of an upper
    limit with NaN for the limit: [.., nan, 0, -1, 0].  Fixture: such a cell and its two
    neighbours from a randomised light curve (tools/make_golden_nan_profile.py)."""
    from godot_amd import core, ops
    dev = _dev()
    g = golden('nan_profile_cell.npz')
    off = np.append(0, np.cumsum(g['counts'])).astype(np.int64)
    cr = ops.CellRanges.from_offsets(torch.from_numpy(g['we']).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(g['S']).to(dev), torch.from_numpy(g['B']).to(dev))
    batch = core.CellBatch(cr.we, cr.we, cr.lo, cr.hi, torch.from_numpy(g['starts']).to(dev),
                           torch.from_numpy(g['stops']).to(dev), cr.S, cr.B)
    assert np.isnan(g['dom'][1]).all() and np.isfinite(g['dom'][[0, 2]]).all()
    for allow in (False, True):
        cll = core.CellsLogLikelihood(batch, allow_failed_cells=allow)      # does not raise
        assert np.isnan(cll._dom[1]).all() and np.isnan(cll._cod[1]).all()
        _close(cll._dom[[0, 2]], g['dom'][[0, 2]])
        lc = cll.get_raw_lightcurve(tsmin=4)
        ref = g['rows']
        assert np.array_equal(np.isnan(lc), np.isnan(ref))
        assert np.array_equal(lc[1, 3:], [0.0, -1.0, 0.0]) and np.isnan(lc[1, 2])
        assert np.array_equal(lc[:, 0], ref[:, 0])
        sc = np.maximum(1.0, np.add.reduceat(np.abs(np.log1p(-g['we'])), off[:-1]))
        for k in (1, 2, 3, 4):
            _close(lc[[0, 2], k], ref[[0, 2], k], rtol=1e-8)
        _close(lc[[0, 2], 5], ref[[0, 2], 5], scale=sc[[0, 2]], rtol=1e-8)


@pytest.mark.parametrize('npt', [2, 3, 5, 63, 64, 65, 100, 200, 255, 256, 257, 300, 1000])
def test_profile_quantiles_match_oracle_at_any_length(npt):
    """Quantiles of the normalised profile (core.py:1247-1256) on random profiles: both kernels
    (rows of at most 256 points in registers, longer ones through LDS), non-uniform abscissae,
    peaks at either end, ties, a row of only NaN, a -inf entry."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(npt)
    ncell = 300
    dom = np.sort(rng.uniform(0, 5, (ncell, npt)), axis=1) + np.arange(npt) * 1e-3
    mu = rng.uniform(-1, 6, (ncell, 1))
    sig = rng.uniform(0.05, 3, (ncell, 1))
    cod = -0.5 * ((dom - mu) / sig) ** 2
    cod[::7] = np.round(cod[::7], 1)          # ties for the arg-max
    cod[5] = np.nan                           # a profile of only NaN
    cod[6, npt // 2] = -np.inf
    for conf in ((0.159, 0.841), (0.05, 0.95), (0.001, 0.999)):
        ag, xlo, xhi = [x.cpu().numpy() for x in ops.profile_quantiles(
            torch.from_numpy(dom).to(dev), torch.from_numpy(cod).to(dev), conf[0], conf[1])]
        for i in range(ncell):
            with np.errstate(all='ignore'):
                ref_ag = dom[i][np.argmax(cod[i])] if not np.isnan(cod[i]).all() else np.nan
                try:
                    ref = orc.flux_quantiles(dom[i], cod[i].copy(), list(conf))
                except IndexError:
                    ref = None                 # searchsorted ran off the end: numpy raises
            if np.isnan(cod[i]).all():
                assert np.isnan(ag[i]) and np.isnan(xlo[i]) and np.isnan(xhi[i])
                continue
            assert ag[i] == ref_ag, (i, ag[i], ref_ag)
            if ref is None:
                continue
            for got, want in ((xlo[i], ref[0]), (xhi[i], ref[1])):
                if np.isfinite(want):
                    assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (npt, conf, i, got, want)


def test_met2mjd_is_the_extended_precision_result_bit_for_bit():
    """met2mjd (core.py:24-26) works in numpy's longdouble (64-bit significand on x86) and
    the light curve stores the result as double: two roundings in extended precision, then
    one to double.  The device restates them in integers; plain double arithmetic differs
    from the reference in about 6 % of the values."""
    from godot_amd import core, ops
    if np.finfo(np.longdouble).nmant != 63:
        pytest.skip('numpy longdouble is not the x87 extended format on this host')
    dev = _dev()
    rng = np.random.default_rng(5)
    t = np.concatenate([2.4e8 + rng.random(300000) * 5e8, rng.random(50000) * 1e3,
                        10.0 ** rng.uniform(-5, 14, 50000),
                        86400. * rng.integers(1, 100000, 50000).astype(float),
                        2.4e8 + np.arange(50000) * 600. + 300])
    ref = np.asarray(core.met2mjd(t), dtype=np.float64)
    got = ops.met2mjd(torch.from_numpy(t).to(dev), core.MJDREF).cpu().numpy()
    assert np.array_equal(got, ref)
    assert (t * (1. / 86400) + core.MJDREF != ref).sum() > 1000       # the comparison has teeth
    # outside the restatement: NaN, never a wrong number
    odd = np.array([0.0, -5.0, np.inf, np.nan, 5e-324, 1e300])
    got = ops.met2mjd(torch.from_numpy(odd).to(dev), core.MJDREF).cpu().numpy()
    assert np.isnan(got).all()


def test_spread_cells_on_two_streams_do_not_interleave():
    """Spread cells exchange partial sums through one device array of the library; launches
    on different streams are chained by the library, so calls issued from two streams at
    once give the bits of calls issued one after the other."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(77)
    sets = []
    for k in range(2):
        cnt = np.array([300001, 250000 + 17 * k, 410003], dtype=np.int64)
        off = np.append(0, np.cumsum(cnt)).astype(np.int64)
        we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
        S = cnt * 0.17 * np.array([0.5, 1.0, 2.0])
        sets.append(ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                                torch.from_numpy(S).to(dev), torch.from_numpy(S * 5).to(dev)))
    serial = []
    for cr in sets:
        a, t, f = ops.cell_fit(cr)
        d, c, g = ops.cell_logpdf(cr, dlogl=20, npt=100)
        serial.append((a, t, d, c))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in sets]
    for rep in range(3):
        got = []
        for cr, st in zip(sets, streams):
            with torch.cuda.stream(st):
                a, t, f = ops.cell_fit(cr)
                d, c, g = ops.cell_logpdf(cr, dlogl=20, npt=100)
                got.append((a, t, d, c, f, g))
        torch.cuda.synchronize()
        for want, have in zip(serial, got):
            for w, h in zip(want, have[:4]):
                assert torch.equal(w, h)
            assert not (have[4].cpu().numpy() & 0x10).any() and not (have[5].cpu().numpy() & 0x10).any()


def test_joint_fit_meets_reference_likelihood(golden):
    """get_max(profile_background=True): the reference's optimiser (scipy TNC,
    ftol 1e-3) defines (alpha, beta) only to its tolerance, so the contract is
    on logL: never lower than the reference's by more than its own margin."""
    from godot_amd import ops
    g = golden('profile_bkg.npz')
    cr, off = _ranges(g, _dev())
    a, b, L, fl = ops.cell_fit_joint(cr)
    a, b, L, fl = a.cpu().numpy(), b.cpu().numpy(), L.cpu().numpy(), fl.cpu().numpy()
    assert not (fl & 0x10).any()
    assert (L >= g['joint_logl'] - 2e-3).all(), (L - g['joint_logl']).min()
    assert (a >= 0).all() and (b >= 0).all()
    # converged: the gradient vanishes, or points outwards on an active bound
    with np.errstate(all='ignore'):
        for i in range(len(a)):
            cl = orc.CellLike(g['we'][off[i]:off[i + 1]], g['exp'][i], g['SonB'][i])
            _, (nga, ngb) = cl.neg_logl_grad(a[i], b[i])
            sc = max(1.0, cl.S, cl.B)
            assert (abs(nga) <= 1e-7 * sc) or (a[i] == 0 and nga >= 0), (i, a[i], nga)
            assert (abs(ngb) <= 1e-7 * sc) or (b[i] == 0 and ngb >= 0), (i, b[i], ngb)
            assert abs(L[i] - cl.log_full_likelihood(a[i], b[i])) <= 1e-9 * max(1, abs(L[i]))
    # distance to the reference's values: informative, bounded by the optimiser tolerance
    assert np.abs(a - g['joint_ab'][:, 0]).max() < 0.2


def test_joint_fit_with_singular_curvature_ends_on_the_edge():
    """One photon, or all weights equal: the Hessian of logL in (alpha, beta)
    has rank one and the maximum is on an edge (alpha = n/S, beta = 0 or
    alpha = 0, beta = n/B).  The cells are fitted, not flagged: the point
    returned meets the Kuhn-Tucker conditions and its logL is the reference's
    (get_max(profile_background=True), core.py:552-630) within its ftol."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(77)
    cells = []
    for n in (1, 1, 1, 2, 5, 40):
        for w0 in (0.03, 0.5, 0.97):
            for s in (0.4, 3.0):
                cells.append((np.full(n, w0), s * n * w0, n * (1 - w0) / s))
    for _ in range(20):                       # one photon, arbitrary exposures
        cells.append((rng.uniform(0.01, 0.99, 1), rng.uniform(0.05, 5), rng.uniform(0.05, 5)))
    we = np.concatenate([c[0] for c in cells])
    off = np.append(0, np.cumsum([len(c[0]) for c in cells])).astype(np.int64)
    S = np.array([c[1] for c in cells])
    B = np.array([c[2] for c in cells])
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    a, b, L, fl = [x.cpu().numpy() for x in ops.cell_fit_joint(cr)]
    assert not (fl & 0x10).any(), np.flatnonzero(fl & 0x10)
    assert (a >= 0).all() and (b >= 0).all()
    with np.errstate(all='ignore'):
        for i in range(len(cells)):
            cl = orc.CellLike(we[off[i]:off[i + 1]], S[i], S[i] / B[i])
            _, (nga, ngb) = cl.neg_logl_grad(a[i], b[i])
            sc = max(1.0, cl.S, cl.B)
            assert (abs(nga) <= 1e-7 * sc) or (a[i] == 0 and nga >= 0), (i, a[i], b[i], nga)
            assert (abs(ngb) <= 1e-7 * sc) or (b[i] == 0 and ngb >= 0), (i, a[i], b[i], ngb)
            assert abs(L[i] - cl.log_full_likelihood(a[i], b[i])) <= 1e-9 * max(1, abs(L[i]))
            try:
                ra, rb = cl.get_max_profile()
            except Exception:
                continue
            assert L[i] >= cl.log_full_likelihood(ra, rb) - 2e-3, (i, a[i], b[i], ra, rb)


def test_joint_fit_moves_on_when_the_coupled_step_is_cut_off_at_a_bound():
    """Two faint photons against a large background exposure: on the way to the interior
    maximum beta comes within 1e-10 of zero, the Newton step is cut off at the bound in
    beta and then points downhill in alpha.  The search goes on with the diagonal step;
    it used to stop there, short of the maximum by 0.76 in logL (found by a randomised
    sweep against the oracle's fmin_tnc fit, core.py:552-630)."""
    from godot_amd import ops
    dev = _dev()
    we = np.array([0.026362088441123657, 0.027453587055668718])
    S, B = np.array([0.11848754739967783]), np.array([4.387810338982554])
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.tensor([0, 2], device=dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    a, b, L, fl = [x.cpu().numpy() for x in ops.cell_fit_joint(cr)]
    assert not (fl & 0x10).any()
    cl = orc.CellLike(we, S[0], S[0] / B[0])
    with np.errstate(all='ignore'):
        ra, rb = cl.get_max_profile()
        assert L[0] >= cl.log_full_likelihood(ra, rb) - 2e-3, (a, b, L, ra, rb)
        _, (nga, ngb) = cl.neg_logl_grad(a[0], b[0])
    sc = max(1.0, cl.S, cl.B)
    assert (abs(nga) <= 1e-7 * sc) or (a[0] == 0 and nga >= 0), (a, nga)
    assert (abs(ngb) <= 1e-7 * sc) or (b[0] == 0 and ngb >= 0), (b, ngb)


def test_profile_background_grid_matches_reference(golden):
    """_get_logpdf_profile centred on a given aopt has no third-party
    optimiser inside (Brent restated): 1e-9 parity."""
    from godot_amd import ops
    g = golden('profile_bkg.npz')
    cr, off = _ranges(g, _dev())
    n = len(g['prof_dom'])
    sub = ops.CellRanges(cr.we, cr.lo[:n].contiguous(), cr.hi[:n].contiguous(), cr.S[:n].contiguous(),
                         cr.B[:n].contiguous())
    aopt = torch.from_numpy(g['joint_ab'][:n, 0].copy()).to(cr.device)
    dom, cod, fl = ops.cell_logpdf_profile(sub, aopt=aopt, dlogl=30, npt=50)
    dom, cod = dom.cpu().numpy(), cod.cpu().numpy()
    ok = ~np.isnan(g['prof_dom'][:, 0])
    assert ok.sum() >= 10
    sc = _logscale(g, off)[:n, None]
    # the support limits are roots found to |dx| ~ 2e-12 of a function with slope O(S)
    _close(dom[ok], g['prof_dom'][ok], rtol=1e-8)
    _close(cod[ok], g['prof_cod'][ok], scale=np.broadcast_to(sc * 10, cod.shape)[ok], rtol=1e-8)


def test_profile_background_lightcurve_close_to_reference(golden):
    from godot_amd import core
    g = golden('profile_bkg.npz')
    d = _dev()
    n = len(g['prof_lightcurve'])
    off = np.append(0, np.cumsum(g['counts'])).astype(np.int64)
    offd = torch.from_numpy(off).to(d)
    sexp = torch.from_numpy(g['exp']).to(d)
    bexp = torch.from_numpy(g['exp'] / g['SonB']).to(d)
    batch = core.CellBatch(torch.from_numpy(g['we']).to(d), torch.from_numpy(g['we']).to(d),
                           offd[:-1][:n].contiguous(), offd[1:][:n].contiguous(),
                           torch.from_numpy(g['starts'][:n].copy()).to(d),
                           torch.from_numpy(g['stops'][:n].copy()).to(d), sexp[:n].contiguous(),
                           bexp[:n].contiguous())
    cll = core.CellsLogLikelihood(batch, profile_background=True)
    lc = cll.get_raw_lightcurve()
    ref = g['prof_lightcurve']
    # profile grids start from the joint maximum (optimiser tolerance): loose comparison
    assert np.abs(cll._dom[:, -1] - g['prof_dom_full'][:, -1]).max() < 0.05 * g['prof_dom_full'][:, -1].max()
    same_kind = (lc[:, 4] == -1) == (ref[:, 4] == -1)
    assert same_kind.mean() > 0.9
    m = same_kind
    assert np.abs(lc[m, 5] - ref[m, 5]).max() < 0.05 + 1e-3 * np.abs(ref[m, 5]).max()
    assert np.abs(lc[m, 2] - ref[m, 2]).max() < 0.05
    assert np.array_equal(lc[:, 0], ref[:, 0]) and np.allclose(lc[:, 1], ref[:, 1], rtol=1e-12)


def test_batch_sanity_check_reports_as_the_reference_does(golden, capsys):
    """CellsLogLikelihood.sanity_check (reference core.py:1100-1106): indices of cells with a
    support narrower than 1e-3 and of cells whose profile does not peak at zero."""
    from godot_amd import core
    g = golden('cells_sparse.npz')
    d = _dev()
    n = 60
    off = np.append(0, np.cumsum(g['counts'])).astype(np.int64)
    offd = torch.from_numpy(off).to(d)
    sexp = torch.from_numpy(g['exp']).to(d)
    bexp = torch.from_numpy(g['exp'] / g['SonB']).to(d)
    batch = core.CellBatch(torch.from_numpy(g['we']).to(d), torch.from_numpy(g['we']).to(d),
                           offd[:-1][:n].contiguous(), offd[1:][:n].contiguous(),
                           torch.from_numpy(g['starts'][:n].copy()).to(d),
                           torch.from_numpy(g['stops'][:n].copy()).to(d), sexp[:n].contiguous(),
                           bexp[:n].contiguous())
    cll = core.CellsLogLikelihood(batch)
    bad_x, bad_y = cll.sanity_check()
    dom, cod = cll._dom, cll._cod
    assert np.array_equal(bad_x, np.ravel(np.argwhere(np.abs(dom[:, -1] - dom[:, 0]) < 1e-3)))
    assert np.array_equal(bad_y, np.ravel(np.argwhere(np.abs(cod.max(axis=1)) > 0.2)))
    out = capsys.readouterr().out
    assert 'suspiciously narrow support' in out and 'substantial disagreement' in out
    # a profile that peaks away from zero and a degenerate support are reported
    cll._cod_h = cod.copy()
    cll._cod_h[7] -= 1.0
    cll._dom_h = dom.copy()
    cll._dom_h[11] = dom[11, 0]
    bad_x, bad_y = cll.sanity_check()
    assert 11 in bad_x and 7 in bad_y


def test_profile_background_with_swap_zeroes_ts_where_the_swapped_exposure_vanishes():
    """get_ts looks at the exposure the swapped likelihood calls S, which is the background
    exposure (reference core.py:343-345, 982-984): a cell without background exposure gets
    TS = 0 under swap, one without source exposure does not."""
    from godot_amd import core
    d = _dev()
    rng = np.random.default_rng(3)
    cnt = np.array([30, 25, 28, 31])
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    we = np.clip(rng.beta(0.6, 1.2, int(off[-1])), 1e-3, 1 - 1e-3)
    sexp = np.array([6.0, 5.0, 5.5, 6.2])
    bexp = np.array([22.0, 0.0, 20.0, 21.0])      # cell 1: no background exposure
    offd = torch.from_numpy(off).to(d)
    wd = torch.from_numpy(we).to(d)
    batch = core.CellBatch(wd, wd, offd[:-1].contiguous(), offd[1:].contiguous(),
                           torch.arange(4, dtype=torch.float64, device=d) * 600. + 3e8,
                           torch.arange(1, 5, dtype=torch.float64, device=d) * 600. + 3e8,
                           torch.from_numpy(sexp).to(d), torch.from_numpy(bexp).to(d))
    cll = core.CellsLogLikelihood.__new__(core.CellsLogLikelihood)
    cll.cells = batch
    cll._swap = True
    cll.profile_background = True
    cll.allow_failed_cells = False
    cll._cr = batch.ranges()
    dom = torch.linspace(0.0, 3.0, 50, dtype=torch.float64, device=d).repeat(4, 1).contiguous()
    cod = (-(dom - 1.0) ** 2).contiguous()
    cll._dom_d, cll._cod_d, cll._dom_h, cll._cod_h = dom, cod, None, None
    aopt, ts, _, _ = cll.get_flux_all()
    ts = ts.cpu().numpy()
    assert ts[1] == 0.0 and (ts[[0, 2, 3]] != 0.0).all()


def test_small_per_cell_functions_match_reference(golden):
    """log_full_likelihood, fmin_tnc_func, f1/f2/f3, f1_profile,
    log_profile_likelihood_approx, nr, halley, the bracketing maxima and
    get_profile_flux of the single-cell API, evaluated through godot_cell_sums."""
    from godot_amd import core
    g = golden('cell_aux.npz')
    alphas, points = g['alphas'], g['points']

    def close(got, ref, what, rtol=RTOL):
        got, ref = np.asarray(got, dtype=float), np.asarray(ref, dtype=float)
        scale = np.maximum(1.0, np.abs(ref))
        with np.errstate(invalid='ignore'):
            ok = (np.abs(got - ref) <= rtol * scale) | (np.isnan(got) & np.isnan(ref))
        assert ok.all(), (what, got, ref)

    for ic in range(int(g['ncase'])):
        for swap in (False, True):
            tag = 'c%d%s_' % (ic, 's' if swap else '')
            tstart, tstop, exp, SonB = g[tag + 'cell']
            we = g[tag + 'we']
            cell = core.Cell(tstart, tstop, exp, np.zeros(len(we)), we, SonB)
            cl = core.CellLogLikelihood(cell, swap=swap)
            n = max(1, len(we))
            close([cl.log_full_likelihood(list(p)) for p in points], g[tag + 'full'], tag + 'full', RTOL * n)
            tn = [cl.fmin_tnc_func(list(p)) for p in points]
            close([[t[0], t[1][0], t[1][1]] for t in tn], g[tag + 'tnc'], tag + 'tnc', RTOL * n)
            close([[cl.f1(a), cl.f2(a), cl.f3(a)] for a in alphas], g[tag + 'f123'], tag + 'f123', RTOL * n)
            close([cl.f1_profile(a) for a in alphas], g[tag + 'f1_profile'], tag + 'f1_profile', 1e-8 * n)
            close([cl.log_profile_likelihood_approx(a) for a in alphas], g[tag + 'lpl_approx'], tag + 'lpl',
                  1e-8 * n)
            close([cl.nr(), cl.nr(guess=2.5, niter=3)], g[tag + 'nr'], tag + 'nr')
            close([cl.halley(), cl.halley(guess=0.4, niter=2)], g[tag + 'halley'], tag + 'halley')
            # brentq stops within xtol = 1e-3: same function to rounding, same iterates
            close([cl.get_max_numerical(), cl.get_max_numerical(guess=3, beta=1.2)], g[tag + 'max_num'],
                  tag + 'max_num', 1e-6)
            close([cl.get_beta_max_numerical(a) for a in (0.5, 1.0, 2.0)], g[tag + 'beta_num'], tag + 'beta_num',
                  1e-6)
            close(cl.get_profile_flux(), g[tag + 'profile_flux'], tag + 'profile_flux', 1e-9 * n)
            if (tag + 'grid') in g:
                ag, bg, rv = cl.get_likelihood_grid(amax=3, bmax=2, res=0.25)
                assert np.array_equal(ag, g[tag + 'grid_a']) and np.array_equal(bg, g[tag + 'grid_b'])
                ref = g[tag + 'grid']
                assert rv.shape == ref.shape
                # (alpha, beta) = (0, 0) is log(0) per photon: -inf here, NaN or -inf in the reference
                # (its t = 1 - w - (1 - w) rounds to +-1e-17); everywhere else the grids agree
                fin = np.isfinite(ref)
                assert np.array_equal(np.isfinite(rv), fin) and not fin[0, 0]
                close(rv[fin], ref[fin], tag + 'grid', RTOL * n)
            # fmin_tnc (ftol 1e-3): an optimiser on nearly the same numbers; its own tolerance applies
            mp = cl.get_max_numerical(profile_background=True)
            ref = g[tag + 'max_num_profile']
            assert cl.log_full_likelihood(list(mp)) >= cl.log_full_likelihood(list(ref)) - 2e-3


def test_flat_layout_likelihood_and_zero_weight_match_reference(golden):
    """CellsLogLikelihoodOld.log_likelihood (one alpha per cell) and Data.zero_weight
    followed by binning."""
    from godot_amd import core
    g = golden('cell_aux.npz')
    dev = _dev()
    data = core.Data(g['in_ti'], g['in_we'], g['in_TSTART'], g['in_TSTOP'], g['in_LIVETIME'], g['in_exposure'],
                     device=dev)
    cells = data.get_cells(tcell=86400., use_barycenter=False, quiet=True)
    n = int(g['old_ncell'])
    old = core.CellsLogLikelihoodOld([cells[i] for i in range(n)], device=dev)
    ll = old.log_likelihood(g['old_alpha'])
    assert abs(ll - float(g['old_ll'])) <= RTOL * abs(float(g['old_ll']))
    a, b = g['zw_range']
    data.zero_weight(a, b)
    S, B, E, nph, nrow = g['zw_totals']
    assert data.ti.numel() == int(nph) and data.TSTART.numel() == int(nrow)
    assert abs(data.S - S) <= 1e-12 * S and abs(data.B - B) <= 1e-12 * B and data.E == E
    zc = data.get_cells(tcell=86400., use_barycenter=False, quiet=True)
    assert np.array_equal(zc.starts.cpu().numpy(), g['zw_starts'])
    assert np.array_equal((zc.hi - zc.lo).cpu().numpy(), g['zw_counts'])
    assert np.allclose(zc.sexp.cpu().numpy(), g['zw_exp'], rtol=1e-12, atol=0)
    dt = data.get_deadtime(torch.tensor([g['zw_starts'][0], g['zw_starts'][-1]], dtype=torch.float64, device=dev))
    assert float(dt[1]) > float(dt[0]) >= 0


@pytest.mark.parametrize('mean', [40, 100, 130, 228, 290, 450, 600, 950, 1200])
def test_dense_team_shapes(mean):
    """One occupancy per register-slot shape of the team kernels between the
    power-of-two ones (8x12 ... 64x24): fit, TS, likelihood, moments and the
    profile grid against the oracle."""
    from godot_amd import ops
    dev = _dev()
    ncell = 260
    rng = np.random.default_rng(mean)
    cnt = rng.poisson(mean, ncell).astype(np.int64)
    cnt[:2] = 0
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
    S = np.maximum(cnt * 0.17 * rng.uniform(0.3, 3, ncell), 0.05)
    B = S * 5
    cr = ops.CellRanges.from_offsets(torch.from_numpy(we).to(dev), torch.from_numpy(off).to(dev),
                                     torch.from_numpy(S).to(dev), torch.from_numpy(B).to(dev))
    alpha, ts, flags = ops.cell_fit(cr, beta=1.3)
    a13, t13 = alpha.cpu().numpy(), ts.cpu().numpy()
    alpha, ts, flags = ops.cell_fit(cr)
    ll = ops.cell_loglike(cr, alpha).cpu().numpy()
    w1, w2 = ops.cell_moments(cr.we, cr.lo, cr.hi)
    dom, cod, lfl = ops.cell_logpdf(cr, dlogl=30, npt=200)
    a, t, dom, cod = alpha.cpu().numpy(), ts.cpu().numpy(), dom.cpu().numpy(), cod.cpu().numpy()
    assert not (flags.cpu().numpy() & 0x10).any() and not (lfl.cpu().numpy() & 0x10).any()
    pick = np.concatenate([[0, 1], rng.choice(np.arange(2, ncell), 40, replace=False)])
    with np.errstate(all='ignore'):
        for c in pick:
            w = we[off[c]:off[c + 1]]
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            ar = cl.get_max()
            assert abs(a[c] - ar) <= RTOL * max(1, abs(ar)), (c, cnt[c])
            assert abs(t[c] - cl.get_ts(ar)) <= RTOL * max(sc, abs(t[c])), (c, cnt[c])
            ab = cl.get_max(beta=1.3)
            assert abs(a13[c] - ab) <= RTOL * max(1, abs(ab)), (c, cnt[c])
            assert abs(ll[c] - cl.log_likelihood(ar)) <= RTOL * max(sc, abs(ll[c])), (c, cnt[c])
            assert abs(float(w1[c]) - w.sum()) <= 1e-12 * max(1.0, w.sum())
            assert abs(float(w2[c]) - (w * w).sum()) <= 1e-12 * max(1.0, w.sum())
        for c in pick[:8]:
            w = we[off[c]:off[c + 1]]
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            d, v = orc.CellLike(w, S[c], S[c] / B[c]).get_logpdf(dlogl=30, npt=200)
            if np.isnan(d).any() or np.isnan(v).any():
                continue
            assert np.abs(dom[c] - d).max() <= RTOL * max(1.0, np.abs(d).max()), (c, cnt[c])
            assert np.abs(cod[c] - v).max() <= RTOL * max(sc, np.abs(v).max()), (c, cnt[c])


def test_a_slice_of_a_cell_list_is_fitted_with_the_shape_of_its_own_cells():
    """A thousand daily cells cut out of a ten-year list share the list's weight array, 10^7
    entries long.  Their photon count travels with them (CellBatch / CellRanges.photons), so
    the kernels size their shape for 2 700 photons a cell, not for array length / cells =
    10^4; the results equal the oracle's, and those of the same cells fitted inside the whole
    list, whichever shape is taken (the shaped entry points change speed only)."""
    from godot_amd import core, ops, synth
    DEV = _dev()
    _n = lambda x: x.cpu().numpy()
    s = synth.synth(10_000_000, years=10.0, seed=11)
    data = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=DEV)
    cells = data.get_cells(tcell=86400., use_barycenter=False, quiet=True)
    whole = cells.ranges()
    a_all, t_all, _ = ops.cell_fit(whole)
    assert cells._photons == whole.photons == int((cells.hi - cells.lo).sum())
    sub = cells[1500:2500]
    assert sub.we.numel() == 10_000_000 and len(sub) == 1000
    cr = sub.ranges()
    mean = cr.photons / cr.ncell
    assert 2000 < mean < 3500                    # inherited from the whole list: no device read
    a, t, fl = ops.cell_fit(cr)
    assert np.abs(_n(a) - _n(a_all)[1500:2500]).max() <= 1e-12
    assert np.abs(_n(t) - _n(t_all)[1500:2500]).max() <= 1e-9 * max(1.0, np.abs(_n(t_all)).max())
    # wrong counts change nothing but the shape: cells taken for tiny, and for huge
    for fake in (1000 * 3, 1000 * 30000):
        cr2 = ops.CellRanges(cr.we, cr.lo, cr.hi, cr.S, cr.B, photons=fake)
        a2, t2, _ = ops.cell_fit(cr2)
        assert np.abs(_n(a2) - _n(a)).max() <= 1e-12
        d2, c2, _ = ops.cell_logpdf(ops.CellRanges(cr.we, cr.lo[:50].contiguous(), cr.hi[:50].contiguous(),
                                                   cr.S[:50].contiguous(), cr.B[:50].contiguous(),
                                                   photons=fake // 20), dlogl=30, npt=200)
        d1, c1, _ = ops.cell_logpdf(ops.CellRanges(cr.we, cr.lo[:50].contiguous(), cr.hi[:50].contiguous(),
                                                   cr.S[:50].contiguous(), cr.B[:50].contiguous()),
                                    dlogl=30, npt=200)
        assert np.abs(_n(d2) - _n(d1)).max() <= 1e-9 * np.abs(_n(d1)).max()
        assert np.abs(_n(c2) - _n(c1)).max() <= 1e-8
    we, lo, hi = s['we'], _n(cr.lo), _n(cr.hi)
    S, B = _n(cr.S), _n(cr.B)
    with np.errstate(all='ignore'):
        for c in range(0, 1000, 25):
            cl = orc.CellLike(we[lo[c]:hi[c]], S[c], S[c] / B[c])
            ar = cl.get_max()
            assert abs(_n(a)[c] - ar) <= 1e-9 * max(1.0, abs(ar))
            assert abs(_n(t)[c] - cl.get_ts(ar)) <= 1e-9 * max(-np.log1p(-we[lo[c]:hi[c]]).sum(), 1.0)


@pytest.mark.gpu
def test_moments_of_sparse_cells_by_a_thread_a_cell_equal_the_teams_to_the_bit():
    """Lists of under three photons a cell take the thread-a-cell kernels (godot_cell_moments,
    godot_cell_series); their additions are those of a team of four lanes: the same cells handed over
    with a weight array that makes the mean look large (so that the team kernels run) give the same bits."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(77)
    ncell = 50000
    cnt = rng.poisson(0.9, ncell).astype(np.int64)
    cnt[rng.choice(ncell, 200, replace=False)] = rng.integers(4, 41, 200)
    cnt[rng.choice(ncell, 20, replace=False)] = rng.integers(33, 300, 20)
    off = np.append(0, np.cumsum(cnt)).astype(np.int64)
    nph = int(off[-1])
    assert nph < 3 * ncell
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    pad = np.concatenate([we, rng.random(5 * ncell - nph)])     # same cells, the mean looks like 5 photons a cell: teams of four lanes
    lo, hi = torch.from_numpy(off[:-1].copy()).to(dev), torch.from_numpy(off[1:].copy()).to(dev)
    w1s, w2s = ops.cell_moments(torch.from_numpy(we).to(dev), lo, hi)
    w1t, w2t = ops.cell_moments(torch.from_numpy(pad).to(dev), lo, hi)
    assert torch.equal(w1s, w1t) and torch.equal(w2s, w2t)
    ref1 = np.array([we[off[c]:off[c + 1]].sum() for c in range(ncell)])
    ref2 = np.array([(we[off[c]:off[c + 1]] ** 2).sum() for c in range(ncell)])
    assert np.allclose(w1s.cpu().numpy(), ref1, rtol=1e-13, atol=0) and np.allclose(w2s.cpu().numpy(), ref2, rtol=1e-13, atol=0)
    exp = rng.random(ncell)
    exp[rng.choice(ncell, 500, replace=False)] = 0.0
    out = []
    for arr in (we, pad):
        e = torch.from_numpy(exp).to(dev)
        se, be = (e * 0.3).clone(), (e * 0.7).clone()
        series = ops.cell_series(torch.from_numpy(arr).to(dev), lo, hi, e, se, be)
        out.append(series + (se, be))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert float(out[0][0][torch.from_numpy(exp == 0).to(dev)].abs().max()) == 0.0


@pytest.mark.gpu
def test_every_mean_occupancy_finds_a_kernel():
    """The shape table of the lane-group kernels over a fine grid of photons a cell: every shape it names
    is one a kernel was built for (the call fails otherwise), for the fit and for the profile grid, and
    the first cell agrees with the oracle."""
    from godot_amd import ops
    dev = _dev()
    rng = np.random.default_rng(21)
    sizes = sorted(set([int(round(x)) for x in np.geomspace(1, 3200, 140)] + list(range(8, 64))))
    for n in sizes:
        ncell = 96
        nph = n * ncell
        we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
        off = (np.arange(ncell + 1) * n).astype(np.int64)
        S = np.full(ncell, max(0.17 * n, 0.05))
        cr = ops.CellRanges(torch.from_numpy(we).to(dev), torch.from_numpy(off[:-1].copy()).to(dev),
                            torch.from_numpy(off[1:].copy()).to(dev), torch.from_numpy(S).to(dev),
                            torch.from_numpy(S * 5).to(dev))
        alpha, ts, flags = ops.cell_fit(cr)
        dom, cod, pf = ops.cell_logpdf(cr, dlogl=30., npt=50)
        cl = orc.CellLike(we[:n], S[0], 0.2)
        with np.errstate(all='ignore'):
            a = cl.get_max()
        assert abs(float(alpha[0].cpu()) - a) <= RTOL * max(1, abs(a)), (n, float(alpha[0].cpu()), a)
        assert bool(torch.isfinite(cod[0]).all().cpu()), n
