"""The hot path at the sizes the headline numbers are quoted on, checked
through properties that do not need the (slow) oracle on every unit:
conservation of photons, the maximum property of the fitted flux, TS from
the likelihood differences, invariance under splitting the job, additivity
of the trig sums, plus oracle spot checks on a few hundred units."""
import numpy as np
import pytest
import torch

from oracle import godot_oracle as orc

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
RTOL = 1e-9


def _n(t):
    return t.cpu().numpy()


@pytest.fixture(scope='module')
def c2():
    """10^7 photons over ten years in 600-s cells (5.26e5 cells, 4.4e5 of them with exposure)."""
    from godot_amd import core, synth
    s = synth.synth(10_000_000, years=10.0, seed=11)
    data = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=DEV)
    cells = data.get_cells(tcell=600.0, use_barycenter=False, quiet=True)
    return s, data, cells


def test_c2_binning_conserves_photons_and_orders_cells(c2):
    s, data, cells = c2
    lo, hi = _n(cells.lo), _n(cells.hi)
    assert len(lo) > 400_000
    assert (hi >= lo).all() and (lo[1:] >= hi[:-1]).all()
    starts, stops = _n(cells.starts), _n(cells.stops)
    assert (stops > starts).all() and (starts[1:] >= stops[:-1]).all()
    assert (_n(cells.sexp) > 0).all() and (_n(cells.bexp) > 0).all()
    # every photon inside a kept cell is in that cell, and only those
    ti = s['ti']
    inside = np.zeros(len(ti), dtype=bool)
    cell_of = np.searchsorted(stops, ti)            # photon on an edge -> earlier cell
    ok = cell_of < len(stops)
    inside[ok] = ti[ok] > starts[cell_of[ok]]
    # the first cell also owns a photon exactly on its start
    assert int((hi - lo).sum()) in (int(inside.sum()), int(inside.sum()) + int((ti == starts[0]).sum()))
    pick = np.random.default_rng(0).choice(len(lo), 2000, replace=False)
    for c in pick:
        t = ti[lo[c]:hi[c]]
        assert ((t > starts[c]) | (c == 0)).all() and (t <= stops[c]).all()
    # summed exposure of the cells = exposure of the span they cover, to rounding
    tot = _n(data.get_exposure(torch.tensor([s['TSTART'][0], stops[-1]], dtype=torch.float64, device=DEV))[1])
    assert abs(_n(cells.sexp).sum() - (tot[1] - tot[0])) <= 1e-9 * tot[1]


def test_c2_fit_is_the_maximum_and_ts_is_the_likelihood_ratio(c2):
    from godot_amd import ops
    s, data, cells = c2
    cr = cells.ranges()
    alpha, ts, flags = ops.cell_fit(cr)
    a, t, fl = _n(alpha), _n(ts), _n(flags)
    assert not (fl & 0x10).any()
    assert np.isfinite(a).all() and (a >= 0).all() and np.isfinite(t).all()
    assert ((fl & 1) != 0).sum() == (a == 0).sum()
    # maximum property: logL(alpha_hat) >= logL(alpha_hat (1 +- 1e-3)) (the reference's own
    # schedule stops within ~1e-6 of the root, far inside the perturbation)
    l0 = _n(ops.cell_loglike(cr, alpha))
    for f in (1 - 1e-3, 1 + 1e-3):
        l1 = _n(ops.cell_loglike(cr, alpha * f))
        assert (l0 - l1 >= -1e-9 * np.maximum(1.0, np.abs(l0)))[a > 0].all()
    # boundary cells: the likelihood falls from alpha = 0
    lz = _n(ops.cell_loglike(cr, torch.zeros_like(alpha)))
    le = _n(ops.cell_loglike(cr, torch.full_like(alpha, 1e-3)))
    assert (lz >= le)[a == 0].all()
    # TS = 2 (logL(alpha_hat) - logL(0)), 0 on the boundary
    cnt = _n(cr.hi - cr.lo)
    scale = np.maximum(1.0, np.abs(lz))
    assert (np.abs(t - 2 * (l0 - lz)) <= RTOL * scale)[a > 0].all()
    assert (t[a == 0] == 0).all()
    # oracle on a few hundred cells, restarted ones included
    rng = np.random.default_rng(5)
    depth = (fl >> 1) & 3
    pick = np.unique(np.concatenate([rng.choice(len(a), 300, replace=False), np.flatnonzero(depth > 0)[:100],
                                     np.argsort(cnt)[-20:]]))
    we = s['we']
    lo, hi, S, B = _n(cr.lo), _n(cr.hi), _n(cr.S), _n(cr.B)
    wdev = _n(cr.we) if cr.we.numel() < 2 ** 27 else None
    with np.errstate(all='ignore'):
        for c in pick:
            w = wdev[lo[c]:hi[c]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            ar = cl.get_max()
            tr = cl.get_ts(ar)
            sc = max(1.0, np.abs(np.log1p(-w)).sum())
            assert abs(a[c] - ar) <= RTOL * max(1, abs(ar)), (c, a[c], ar)
            assert abs(t[c] - tr) <= RTOL * max(sc, abs(tr)), (c, t[c], tr)


def test_c2_every_cell_matches_the_oracle(c2):
    """All 4.4e5 cells of the headline workload against the oracle's get_max + get_ts
    (about 20 s of CPU), not a sample."""
    from godot_amd import ops
    s, data, cells = c2
    cr = cells.ranges()
    alpha, ts, flags = ops.cell_fit(cr)
    a, t, fl = _n(alpha), _n(ts), _n(flags)
    we, lo, hi, S, B = _n(cr.we), _n(cr.lo), _n(cr.hi), _n(cr.S), _n(cr.B)
    ref_a = np.empty(len(a))
    ref_t = np.empty(len(a))
    scale = np.empty(len(a))
    with np.errstate(all='ignore'):
        for c in range(len(a)):
            w = we[lo[c]:hi[c]]
            cl = orc.CellLike(w, S[c], S[c] / B[c])
            ref_a[c] = cl.get_max()
            ref_t[c] = cl.get_ts(ref_a[c])
            scale[c] = max(1.0, -np.log1p(-w).sum())
    assert (np.abs(a - ref_a) <= RTOL * np.maximum(1.0, np.abs(ref_a))).all()
    assert (np.abs(t - ref_t) <= RTOL * np.maximum(scale, np.abs(ref_t))).all()
    assert np.array_equal((fl & 1) != 0, ref_a == 0)


def test_c2_fit_does_not_depend_on_how_the_job_is_split(c2):
    """Cell ranges fitted on their own (what a rank does with its share of the cells, over
    the same weight array) reproduce the whole job: no cell sees its neighbours.  The kernel
    shape follows the shard's occupancy, so sums may be grouped differently:
    agreement to 1e-12, and to the bit for all but a few cells."""
    from godot_amd import ops
    from godot_amd import dist as gd
    s, data, cells = c2
    cr = cells.ranges()
    alpha, ts, flags = ops.cell_fit(cr)
    counts = _n(cr.hi - cr.lo)
    a, t = _n(alpha), _n(ts)
    for world in (2, 8):
        parts = []
        for r in range(world):
            c0, c1 = gd.cell_shard(counts, r, world)
            sub = cr.shard(c0, c1, photons=int(counts[c0:c1].sum()))
            assert sub.nph == cr.nph and sub.photons == int(counts[c0:c1].sum())
            parts.append(ops.cell_fit(sub))
        a2 = _n(torch.cat([p[0] for p in parts]))
        t2 = _n(torch.cat([p[1] for p in parts]))
        f2 = _n(torch.cat([p[2] for p in parts]))
        assert np.abs(a2 - a).max() <= 1e-12 * max(1.0, np.abs(a).max())
        assert (np.abs(t2 - t) <= 1e-10 * np.maximum(1.0, np.abs(t))).all()
        assert ((f2 & 1) == (_n(flags) & 1)).all()
        assert (a2 == a).mean() > 0.99


def test_c2_profile_grid_brackets_the_maximum(c2):
    """a6 at full size: every profile peaks at 0 (within the reference's 0.05
    acceptance), is 0 at the fitted flux, and falls to about -dlogl at its ends."""
    from godot_amd import ops
    s, data, cells = c2
    cr = cells.ranges()
    alpha, _, _ = ops.cell_fit(cr)
    n = 200_000                         # 200k cells x 200 points x 2 arrays = 640 MB
    sub = ops.CellRanges(cr.we, cr.lo[:n].contiguous(), cr.hi[:n].contiguous(), cr.S[:n].contiguous(),
                         cr.B[:n].contiguous())
    dom, cod, fl = ops.cell_logpdf(sub, dlogl=30, npt=200)
    assert not (_n(fl) & 0x10).any()
    top = cod.max(dim=1).values
    assert bool((top.abs() <= 0.05).all())
    assert bool((dom[:, 1:] > dom[:, :-1]).all()) and bool((dom[:, 0] >= 0).all())
    a = alpha[:n]
    inside = (a >= dom[:, 0]) & (a <= dom[:, -1])
    assert bool(inside.all())
    # right end: the search stops within 0.1 of -dlogl (reference core.py:869-916) after at most
    # 5 steps; nearly all cells get there
    right = _n(cod[:, -1])
    assert np.median(np.abs(right + 30)) < 0.1 and (right < -20).mean() > 0.999
    # spot rows against the oracle's get_logpdf: every 500th cell, abscissae and log-densities
    pick = np.arange(0, n, 500)
    dom_h, cod_h = _n(dom[pick]), _n(cod[pick])
    we, lo, hi, S, B = _n(cr.we), _n(cr.lo), _n(cr.hi), _n(cr.S), _n(cr.B)
    with np.errstate(all='ignore'):
        for row, c in enumerate(pick):
            cl = orc.CellLike(we[lo[c]:hi[c]], S[c], S[c] / B[c])
            rd, rc = cl.get_logpdf(npt=200, dlogl=30)
            assert np.abs(dom_h[row] - rd).max() <= RTOL * max(1.0, np.abs(rd).max()), c
            assert np.abs(cod_h[row] - rc).max() <= RTOL * max(30.0, np.abs(rc).max()), c


def test_c2_light_curve_reads_half_of_its_profile_rows(c2):
    """get_flux_all / get_raw_lightcurve take the quantiles from the log-densities and the two
    ends of every row of abscissae (godot_profile_quantiles_linspace): to the bit what
    godot_profile_quantiles gives on the whole rows, for the rows of the fixed-background and of
    the background-profile kernels (NaN rows of cells without a profile included); a row that is
    not numpy.linspace of its ends is told apart by the general kernel only; and spot rows of
    the light curve equal the oracle's raw_lightcurve."""
    from godot_amd import core, ops
    s, data, cells = c2
    n = 150_000
    sub = cells[:n]
    cll = core.CellsLogLikelihood(sub)
    assert cll._rows_linspace
    dom, cod = cll._dom_d, cll._cod_d
    same = lambda a, b: torch.equal(a.view(torch.int64), b.view(torch.int64))
    for lv in ((0.05, 0.95), (0.159, 0.841), (0.5, 0.5)):
        got = ops.profile_quantiles(dom, cod, lv[0], lv[1], linspace=True)
        ref = ops.profile_quantiles(dom, cod, lv[0], lv[1])
        assert all(same(g, r) for g, r in zip(got, ref)), lv
    pd, pc, _ = ops.cell_logpdf_profile(cells[:2000].ranges(), dlogl=20, npt=100)
    pd[7] = float('nan')
    pc[7] = float('nan')
    got = ops.profile_quantiles(pd, pc, 0.16, 0.84, linspace=True)
    ref = ops.profile_quantiles(pd, pc, 0.16, 0.84)
    assert all(same(g, r) for g, r in zip(got, ref))
    bent = dom[:1000].clone()
    bent[:, 100] = 0.5 * (bent[:, 100] + bent[:, 101])
    assert not torch.equal(ops.profile_quantiles(bent, cod[:1000], 0.05, 0.95)[1],
                           ops.profile_quantiles(bent, cod[:1000], 0.05, 0.95, linspace=True)[1])
    lc = cll.get_raw_lightcurve()
    pick = np.arange(0, n, 1500)
    dom_h, cod_h = _n(dom[pick]), _n(cod[pick])
    we, lo, hi, S, B = (_n(x) for x in (sub.we, sub.lo, sub.hi, sub.sexp, sub.bexp))
    with np.errstate(all='ignore'):
        clls = [orc.CellLike(we[lo[c]:hi[c]], S[c], S[c] / B[c]) for c in pick]
        rdom, rcod = orc.cells_logpdf(clls, npt=200, dlogl=30)
        ref = orc.raw_lightcurve(clls, _n(sub.starts)[pick], _n(sub.stops)[pick], rdom, rcod)
    got = lc[pick]
    assert np.array_equal(got[:, 4] == -1, ref[:, 4] == -1)
    scale = np.maximum(1.0, np.abs(ref))
    assert (np.abs(got - ref) <= 1e-7 * scale).all(), np.abs((got - ref) / scale).max()


def test_c4_trig_sums_add_and_match_the_oracle_at_spot_frequencies():
    """10^6 photons x 2^21 frequencies: sums of photon shards add up to the whole
    search; 12 random frequencies against the oracle's exactly reduced phases."""
    from godot_amd import ops, synth
    s = synth.synth(1_000_000, years=10.0, seed=17)
    t, w = s['ti'] - s['ti'][0], s['we']
    td, wd = torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV)
    nf = 1 << 21
    f0, df = 1e-4, 1.0 / (4 * (t[-1] - t[0]))
    z, h = ops.z2m(td, wd, f0, df, nf, m=2, want_h=True)
    zn = _n(z)
    assert np.isfinite(zn).all() and (zn >= 0).all()
    assert bool((h <= z).all()) and bool((h >= z - 4).all())      # H = max(Z1, Z2 - 4)
    js = np.random.default_rng(3).choice(nf, 12, replace=False)
    for j in js:
        ref = orc.z2m(t, w, f0, df, 1, m=2, j0=int(j))[0]
        assert abs(zn[j] - ref) <= 1e-10 * max(1.0, ref), (j, zn[j], ref)
    # two photon shards, one frequency tile
    j0, n = 777_000, 4096
    cut = 400_000
    s1, q1 = ops.z2m_sums(td[:cut].contiguous(), wd[:cut].contiguous(), f0, df, n, m=2, j0=j0)
    s2, q2 = ops.z2m_sums(td[cut:].contiguous(), wd[cut:].contiguous(), f0, df, n, m=2, j0=j0)
    zz, _ = ops.z2m_finish(s1 + s2, q1 + q2)
    assert np.abs(_n(zz) - zn[j0:j0 + n]).max() <= 1e-10 * max(1.0, zn[j0:j0 + n].max())


def test_c4_search_of_all_trial_frequencies_keeps_the_best_ten():
    """C4 at full size: 10^6 photons x 10^8 trial frequencies, m = 2, in 48 launches of 2^21
    frequencies (about half a minute), the global ten best by dist.search_topk_sharded on one
    rank.  The statistic at those ten frequencies against the oracle's exactly reduced phases;
    the list is ordered and complete (no value of the spectrum outside it exceeds its last);
    and two frequency shards, each keeping its own ten best, merge to the same list by the
    rule the ranks use."""
    from godot_amd import ops, synth
    from godot_amd import dist as gd
    s = synth.synth(1_000_000, years=1.0, seed=17)
    t, w = s['ti'] - s['ti'][0], s['we']
    td, wd = torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV)
    nf, chunk = 100_000_000, 1 << 21
    f0, df = 1e-4, 1.0 / (4 * float(t[-1]))
    ws = torch.empty(ops._lib.lib.godot_z2m_workspace_bytes(td.numel(), chunk, 2), dtype=torch.uint8, device=DEV)
    zbuf = torch.empty(nf, dtype=torch.float64, device=DEV)

    def search_fn(j0, n):
        for k0 in range(0, n, chunk):
            nn = min(chunk, n - k0)
            zbuf[k0:k0 + nn] = ops.z2m(td, wd, f0, df, nn, m=2, j0=j0 + k0, workspace=ws)[0]
        return zbuf[:n]

    top_v, top_i = gd.search_topk_sharded(nf, search_fn, k=10)
    assert len(top_v) == 10 and (np.diff(top_v) <= 0).all() and len(set(top_i.tolist())) == 10
    for v, j in zip(top_v, top_i):
        ref = orc.z2m(t, w, f0, df, 1, m=2, j0=int(j))[0]
        assert abs(v - ref) <= 1e-10 * max(1.0, ref), (j, v, ref)
    assert int((zbuf > top_v[-1]).sum()) == 9 and float(zbuf.max()) == top_v[0]
    assert bool(torch.isfinite(zbuf).all()) and float(zbuf.min()) >= 0
    # two shards: rank r's ten best of its slice, gathered and reduced as search_topk_sharded does
    rec = []
    for r in range(2):
        j0, n = gd.frequency_shard(nf, r, 2)
        v, i = torch.topk(zbuf[j0:j0 + n], 10)
        rec.append(np.stack([_n(v), _n(i + j0).astype(np.float64)], axis=1))
    rec = np.concatenate(rec)
    order = np.lexsort((rec[:, 1], -rec[:, 0]))[:10]
    assert np.array_equal(rec[order, 1].astype(np.int64), top_i) and np.array_equal(rec[order, 0], top_v)


def test_c3_power_spectrum_matches_numpy_transforms():
    """C3: twelve years of 100-s bins (3 786 912 of them, padded to l = 2^25): the
    device spectra against the oracle's five numpy transforms of the same series."""
    from godot_amd import core, synth
    s = synth.synth(2_000_000, years=12.0, seed=13)
    data = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=DEV)
    ts = data.get_cells(tcell=100.0, time_series_only=True, trim_zero_exposure=False, use_barycenter=False,
                        quiet=True)
    assert len(_n(ts.starts)) > 3_700_000
    f, p0, p1, pb = core.power_spectrum_fft(ts)
    assert len(f) == (1 << 23) + 1
    with np.errstate(all='ignore'):
        rf, r0, r1, rb = orc.power_spectrum_fft(_n(ts.starts), _n(ts.stops), _n(ts.counts), _n(ts.weights),
                                                _n(ts.weights2), _n(ts.sexp), _n(ts.bexp))
    assert np.array_equal(f, rf)
    for got, ref, nm in ((p0, r0, 'P0'), (p1, r1, 'P1'), (pb, rb, 'Pb')):
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), fin), nm
        tol = RTOL * np.maximum(np.abs(ref[fin]), 1e-3 * np.abs(ref[fin]).max())
        assert (np.abs(got[fin] - ref[fin]) <= tol).all(), (nm, np.abs(got[fin] - ref[fin]).max())


def test_multi_source_batch_matches_the_oracle_cell_for_cell():
    """Eight sources of 1e5 photons (the shape of the multi-source job at a ten-thousandth of
    its size): one batch, one selection, one fit launch; every cell of every source against
    the oracle's binning, get_max and get_ts; and the sources dealt over two ranks
    (dist.source_shard) give the same numbers as the single batch."""
    from godot_amd import core, ops, synth
    from godot_amd import dist as gd
    dev = torch.device('cuda:0')
    nsrc, tcell = 8, 3600.
    raw = [synth.synth(100_000, years=1.0, seed=1000 + k) for k in range(nsrc)]
    for k, s in enumerate(raw):                 # sources differ in brightness, not only in photons
        s['S'] *= 1.0 + 0.1 * k
    datas = []
    for s in raw:
        d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=dev)
        d.S, d.B = s['S'], s['B']               # the totals are attributes, as in the reference (core.py:1583-1585)
        datas.append(d)
    batch = core.get_cells_batch(datas, tcell=tcell)
    assert len(batch) == nsrc
    alpha, ts, flags = ops.cell_fit(batch.cells.ranges())
    assert not (flags.cpu().numpy() & 0x10).any()
    alpha, ts = alpha.cpu().numpy(), ts.cpu().numpy()
    lo, hi = batch.cells.lo.cpu().numpy(), batch.cells.hi.cpu().numpy()
    ncmp = 0
    with np.errstate(all='ignore'):
        for k, s in enumerate(raw):
            b = orc.bin_cells(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['exposure'], s['LIVETIME'], s['S'],
                              s['B'], s['E'], tcell=tcell)
            c0, c1 = int(batch.cell_offsets[k]), int(batch.cell_offsets[k + 1])
            assert c1 - c0 == len(b['counts'])
            assert np.array_equal(hi[c0:c1] - lo[c0:c1], b['counts'])
            assert np.array_equal(batch.cells.starts[c0:c1].cpu().numpy(), b['starts'])
            assert np.array_equal(batch.cells.sexp[c0:c1].cpu().numpy(), b['sexp'])
            for i, cl in enumerate(orc.cells_from_bins(b)):
                a = cl.get_max()
                t = cl.get_ts(a)
                sc = max(1.0, np.abs(np.log1p(-cl.we)).sum())
                assert abs(alpha[c0 + i] - a) <= 1e-9 * max(1, abs(a)), (k, i, alpha[c0 + i], a)
                assert abs(ts[c0 + i] - t) <= 1e-9 * max(sc, abs(t)), (k, i, ts[c0 + i], t)
                ncmp += 1
    assert ncmp == len(alpha) and ncmp > 30000
    # two ranks' worth of sources
    pieces = batch.split(alpha)
    for rank in range(2):
        mine = gd.source_shard(nsrc, rank, 2)
        sub = core.get_cells_batch([datas[k] for k in mine], tcell=tcell)
        a2, _, _ = ops.cell_fit(sub.cells.ranges())
        for got, k in zip(sub.split(a2.cpu().numpy()), mine):
            assert np.allclose(got, pieces[k], rtol=1e-12, atol=0, equal_nan=True)


def test_run_sources_fits_and_spectra_match_the_oracle_source_by_source():
    """The multi-source job through its driver (core.run_sources: compact exposure tables of all
    sources, cells of all sources binned and fitted together, time series binned in chunks,
    one power spectrum per source), four sources of 2e5 photons over two years: every daily
    cell's maximum and TS against the oracle's get_max / get_ts, and every source's P_0, P_1,
    P_b at 300-s bins against the oracle's own binning (bin_cells, zero_low_exposure) and its
    five numpy transforms (power_spectrum_fft), peak and peak frequency included."""
    from godot_amd import core, synth
    dev = torch.device('cuda:0')
    nsrc = 4
    raw = [synth.synth(200_000, years=2.0, seed=1000 + k) for k in range(nsrc)]
    for k, s in enumerate(raw):
        s['S'] *= 1.0 + 0.1 * k
    datas = []
    for s in raw:
        d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=dev)
        d.S, d.B = s['S'], s['B']
        datas.append(d)
    res = core.run_sources(datas, tcell_fit=86400., tcell_ps=300., chunk=3, keep_spectra=True, compact_tables=True)
    assert all(d._exposure_table is None and d._checkpoints is not None for d in datas)
    assert len(res.batch) == nsrc and len(res.spectra) == nsrc
    assert not (res.flags.cpu().numpy() & 0x10).any()
    alpha, ts = res.alpha.cpu().numpy(), res.ts.cpu().numpy()
    with np.errstate(all='ignore'):
        for k, s in enumerate(raw):
            # the fit half
            b = orc.bin_cells(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['exposure'], s['LIVETIME'], s['S'],
                              s['B'], s['E'], tcell=86400.)
            c0, c1 = int(res.batch.cell_offsets[k]), int(res.batch.cell_offsets[k + 1])
            assert c1 - c0 == len(b['counts']) > 700
            for i, cl in enumerate(orc.cells_from_bins(b)):
                a = cl.get_max()
                t = cl.get_ts(a)
                sc = max(1.0, np.abs(np.log1p(-cl.we)).sum())
                assert abs(alpha[c0 + i] - a) <= 1e-9 * max(1, abs(a)), (k, i, alpha[c0 + i], a)
                assert abs(ts[c0 + i] - t) <= 1e-9 * max(sc, abs(t)), (k, i, ts[c0 + i], t)
            # the periodogram half
            b = orc.bin_cells(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['exposure'], s['LIVETIME'], s['S'],
                              s['B'], s['E'], tcell=300., trim_zero_exposure=False)
            b['counts'] = b['counts'].astype(np.float64)
            orc.zero_low_exposure(b, 0)
            rf, r0, r1, rb = orc.power_spectrum_fft(b['starts'], b['stops'], b['counts'], b['wsum'], b['w2sum'],
                                                    b['sexp'], b['bexp'])
            l, tsamp, p0, p1, pb = res.spectra[k]
            assert l == 2 * (len(rf) - 1) * 2 and tsamp == b['stops'][0] - b['starts'][0]
            for j, (got, ref, nm) in enumerate(((p0, r0, 'P0'), (p1, r1, 'P1'), (pb, rb, 'Pb'))):
                got = got.cpu().numpy()
                fin = np.isfinite(ref)
                assert np.array_equal(np.isfinite(got), fin), (k, nm)
                tol = RTOL * np.maximum(np.abs(ref[fin]), 1e-3 * np.abs(ref[fin]).max())
                assert (np.abs(got[fin] - ref[fin]) <= tol).all(), (k, nm, np.abs(got[fin] - ref[fin]).max())
                at = 1 + int(np.nanargmax(ref[1:]))
                assert abs(res.peak_power[k, j] - ref[at]) <= 1e-9 * abs(ref[at]), (k, nm)
                assert res.peak_freq[k, j] == rf[at], (k, nm, res.peak_freq[k, j], rf[at])
    # the same job without keeping the spectra, full tables, one source a chunk: the same summaries
    fresh = []
    for s in raw:
        d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=dev)
        d.S, d.B = s['S'], s['B']
        fresh.append(d)
    res2 = core.run_sources(fresh, tcell_fit=86400., tcell_ps=300., chunk=1, compact_tables=False)
    assert res2.spectra is None and all(d._exposure_table is not None for d in fresh)
    assert np.array_equal(res2.peak_power, res.peak_power) and np.array_equal(res2.peak_freq, res.peak_freq)
    assert torch.equal(res2.alpha, res.alpha)


def test_c1_raw_light_curve_at_full_size():
    """C1 as stated: 1e6 photons (seed 7), ten years, one-day cells; binning, profiles and the
    raw light curve of all 3 652 cells against the oracle."""
    from godot_amd import core, synth
    s = synth.synth(1_000_000, years=10.0, seed=7)
    d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'],
                  device=torch.device('cuda:0'))
    cells = d.get_cells(tcell=86400., use_barycenter=False, quiet=True)
    lc = core.CellsLogLikelihood(cells).get_raw_lightcurve()
    b = orc.bin_cells(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['exposure'], s['LIVETIME'], s['S'], s['B'],
                      s['E'], tcell=86400.)
    assert len(cells) == len(b['counts']) and len(cells) >= 3652
    assert np.array_equal((cells.hi - cells.lo).cpu().numpy(), b['counts'])
    clls = orc.cells_from_bins(b)
    with np.errstate(all='ignore'):
        dom, cod = orc.cells_logpdf(clls)
        ref = orc.raw_lightcurve(clls, b['starts'], b['stops'], dom, cod)
    scale = np.maximum(np.abs(ref), 1.0)
    scale[:, 5] = np.maximum(scale[:, 5], [np.abs(np.log1p(-c.we)).sum() for c in clls])
    assert np.nanmax(np.abs(lc - ref) / scale) < 1e-9
    assert np.array_equal(np.isnan(lc), np.isnan(ref))
