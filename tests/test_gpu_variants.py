"""get_cells variants (flux scalers, randomised deal, cells from time
intervals / contiguous exposures) against the reference's golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _data(g):
    from godot_amd import core
    return core.Data(g['in_ti'], g['in_we'], g['in_TSTART'], g['in_TSTOP'], g['in_LIVETIME'], g['in_exposure'],
                     device=DEV)


def _scalers(g):
    from godot_amd.scalers import PiecewiseScaler
    return (PiecewiseScaler(g['s_starts'], g['s_stops'], g['s_scales']),
            PiecewiseScaler(g['b_starts'], g['b_stops'], g['b_scales']))


def _describe(batch):
    """Same order-independent per-cell description as the golden generator."""
    lo, hi = batch.lo.cpu().numpy(), batch.hi.cpu().numpy()
    we, ti = batch.we.cpu().numpy(), batch.ti.cpu().numpy()
    st = batch.starts.cpu().numpy()
    return dict(starts=st, stops=batch.stops.cpu().numpy(), exp=batch.sexp.cpu().numpy(),
                SonB=(batch.sexp / batch.bexp).cpu().numpy(), counts=hi - lo,
                wsum=np.array([np.sum(np.sort(we[a:b])) for a, b in zip(lo, hi)]),
                w2sum=np.array([np.sum(np.sort(we[a:b]) ** 2) for a, b in zip(lo, hi)]),
                tsum=np.array([np.sum(np.sort(ti[a:b] - s)) for a, b, s in zip(lo, hi, st)]))


def _same(d, g, prefix, exact=('starts', 'stops', 'counts'), rtol=1e-12):
    for k, v in d.items():
        ref = g[prefix + k]
        if k in exact:
            assert np.array_equal(v, ref), (prefix, k)
        else:
            assert np.allclose(v, ref, rtol=rtol, atol=0), (prefix, k, np.abs(v - ref).max())


def test_scaler_values_match_reference(golden):
    g = golden('cell_variants.npz')
    src, _ = _scalers(g)
    assert np.array_equal(src(g['in_TSTART'], g['in_TSTOP']), g['scaler_rows'])
    assert np.array_equal(src(g['in_ti']), g['scaler_events'])
    on_dev = src(torch.from_numpy(g['in_ti']).to(DEV)).cpu().numpy()
    assert np.array_equal(on_dev, g['scaler_events'])
    # averages over the rows of the spacecraft table, on the device: the same bits
    rows = src(torch.from_numpy(g['in_TSTART']).to(DEV), torch.from_numpy(g['in_TSTOP']).to(DEV))
    assert rows.is_cuda and np.array_equal(rows.cpu().numpy(), g['scaler_rows'])


def test_scaler_averages_on_the_device_equal_the_host_arithmetic():
    """Intervals inside one segment, across one boundary, across many, beginning exactly on
    a boundary, before the first and behind the last segment: host and device agree bit for bit."""
    from godot_amd.scalers import PiecewiseScaler
    rng = np.random.default_rng(21)
    edges = np.cumsum(rng.uniform(50., 5000., 41)) + 1e8
    sc = PiecewiseScaler(edges[:-1], edges[1:], rng.uniform(0.2, 5.0, 40))
    t0 = rng.uniform(edges[0] - 5e3, edges[-1] + 5e3, 20000)
    t1 = t0 + rng.choice([30., 300., 3000., 30000.], 20000) * rng.uniform(0.1, 1.0, 20000)
    t0[:40] = edges[:40]                      # starts on the boundaries themselves
    t1[:40] = edges[:40] + rng.uniform(1., 9000., 40)
    host = sc(t0, t1)
    dev = sc(torch.from_numpy(t0).to(DEV), torch.from_numpy(t1).to(DEV)).cpu().numpy()
    assert np.array_equal(host, dev)
    assert (host != sc(t0)).any()             # some intervals did span segments


def test_scaled_cells_match_reference(golden):
    g = golden('cell_variants.npz')
    data = _data(g)
    src, bkg = _scalers(g)
    we_before = data.we
    cells = data.get_cells(tcell=3600., use_barycenter=False, src_scaler=src, bkg_scaler=bkg, quiet=True)
    assert data.we is we_before and data.sexposure is None      # reverted, as in the reference
    d = _describe(cells)
    _same(d, g, 'scaled_', exact=('starts', 'stops', 'counts', 'exp'))
    lo, hi = cells.lo.cpu().numpy(), cells.hi.cpu().numpy()
    we = np.concatenate([cells.we.cpu().numpy()[a:b] for a, b in zip(lo, hi)])
    assert np.array_equal(we, g['scaled_we'])
    ts = data.get_cells(tcell=3600., use_barycenter=False, src_scaler=src, time_series_only=True, quiet=True)
    assert np.array_equal(ts.sexp.cpu().numpy(), g['scaled_ts_sexp'])
    assert np.array_equal(ts.bexp.cpu().numpy(), g['scaled_ts_bexp'])
    assert np.array_equal(ts.counts.cpu().numpy(), g['scaled_ts_counts'])
    assert np.allclose(ts.weights.cpu().numpy(), g['scaled_ts_weights'], rtol=1e-12, atol=1e-13)
    assert np.allclose(ts.weights2.cpu().numpy(), g['scaled_ts_weights2'], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('prefix,seed,scaled', [('rand_', 3, False), ('randscaled_', 4, True)])
def test_randomised_deal_matches_reference(golden, prefix, seed, scaled):
    """Same random stream (numpy default_rng on the host), same insertion points:
    every cell receives the reference's photons (order inside a cell aside)."""
    g = golden('cell_variants.npz')
    data = _data(g)
    kw = dict(src_scaler=_scalers(g)[0]) if scaled else {}
    cells = data.get_cells(tcell=3600., use_barycenter=False, randomize=True, seed=seed, quiet=True, **kw)
    _same(_describe(cells), g, prefix, exact=('starts', 'stops', 'counts', 'exp'))


def test_contiguous_exposure_cells_match_reference(golden):
    g = golden('cell_variants.npz')
    data = _data(g)
    gs, ge = data.get_contiguous_exposures()
    assert np.array_equal(gs, g['contig_starts']) and np.array_equal(ge, g['contig_stops'])
    t0 = g['in_TSTART'][0]
    gs2, ge2 = data.get_contiguous_exposures(tstart=t0 + 1.2345 * 86400, tstop=t0 + 9.8765 * 86400)
    assert np.array_equal(gs2, g['contig2_starts']) and np.array_equal(ge2, g['contig2_stops'])
    cells = data.get_contiguous_exposure_cells()
    _same(_describe(cells), g, 'contig_', exact=('starts', 'stops', 'counts', 'exp'))
    cells = data.get_contiguous_exposure_cells(randomize=True, seed=5)
    _same(_describe(cells), g, 'contigrand_', exact=('starts', 'stops', 'counts', 'exp'), rtol=1e-9)


def test_search_sorted_matches_numpy():
    from godot_amd import ops
    rng = np.random.default_rng(2)
    a = np.sort(rng.integers(0, 50, 1000).astype(float))
    v = np.concatenate([rng.uniform(-5, 55, 500), a[::7], [-1e9, 1e9]])
    for side in ('left', 'right'):
        out = ops.search_sorted(torch.from_numpy(a).to(DEV), torch.from_numpy(v).to(DEV), side=side)
        assert np.array_equal(out.cpu().numpy(), np.searchsorted(a, v, side=side))


def test_phase_cells_and_folded_bayesian_blocks_match_reference(golden):
    """PhaseData.get_cells (plain, converse, randomised) and Bayesian Blocks on the
    folded light curve wrapped three times (PhaseCellsLogLikelihood)."""
    from godot_amd import core
    g = golden('phase_cells.npz')
    pd = core.PhaseData(g['phases'], g['weights'], device=DEV)
    cells = pd.get_cells(ncell=50)
    _same(_describe(cells), g, 'plain_', exact=('starts', 'stops', 'counts', 'exp'))
    _same(_describe(pd.get_cells(ncell=20, get_converse=True)), g, 'converse_',
          exact=('starts', 'stops', 'counts', 'exp'))
    _same(_describe(pd.get_cells(ncell=50, randomize=True, seed=9)), g, 'rand_',
          exact=('starts', 'stops', 'counts', 'exp'))
    assert isinstance(cells[3], core.PhaseCell)
    pcl = core.PhaseCellsLogLikelihood(cells)
    rows, idx = pcl.get_bb_lightcurve(tsmin=4, bb_prior=8, get_indices=True)
    assert np.array_equal(idx, g['bb_idx'])
    ref = g['bb_rows']
    assert np.array_equal(rows[:, 4] == -1, ref[:, 4] == -1)
    tol = 1e-9 * np.maximum(np.abs(ref), 1.0)
    tol[:, 5] *= np.maximum(1.0, np.abs(ref[:, 5]))
    assert (np.abs(rows - ref) <= tol).all(), np.abs(rows - ref).max(0)
    raw = core.CellsLogLikelihood(cells).get_raw_lightcurve()
    rr = g['raw_rows']
    assert np.array_equal(raw[:, 4] == -1, rr[:, 4] == -1)
    tol = 1e-9 * np.maximum(np.abs(rr), 1.0)
    tol[:, 5] *= np.maximum(1.0, np.abs(rr[:, 5]))
    assert (np.abs(raw - rr) <= tol).all()


def test_bb_prior_tune_matches_reference(golden):
    from godot_amd import core
    g = golden('bb_prior_tune.npz')
    data = _data(g)
    priors = [int(x) for x in g['priors']]
    r1, n1 = core.bb_prior_tune(data, 7200., bb_priors=priors, ntrial=2, use_barycenter=False, seed=7, quiet=True)
    assert n1 == int(g['tune_ncell']) and np.array_equal(r1, g['tune_cells'])
    r2, n2 = core.bb_prior_tune(data, None, bb_priors=priors, ntrial=1, orbital=True, seed=8)
    assert n2 == int(g['tune_orbital_ncell']) and np.array_equal(r2, g['tune_orbital'])


def _bary_converters(g):
    """The analytic converters the golden generator handed to the reference."""
    amp, period, t0 = float(g['bary_amp']), float(g['bary_period']), g['in_TSTART'][0]

    def t2b(t):
        t = np.asarray(t, dtype=np.float64)
        return t + amp * np.sin(2 * np.pi * (t - t0) / period)

    def b2t(b):
        b = np.asarray(b, dtype=np.float64)
        t = b.copy()
        for _ in range(6):
            t = b - amp * np.sin(2 * np.pi * (t - t0) / period)
        return t
    return t2b, b2t


@pytest.mark.parametrize('timeref,tag', [('LOCAL', 'topo'), ('SOLARSYSTEM', 'ssb')])
def test_barycentred_cells_match_reference(golden, timeref, tag):
    """use_barycenter=True with caller-supplied time converters: uniform cells at
    the barycentre, exposure (and, for topocentric photon times, binning) on the
    converted edges."""
    from godot_amd import core
    g = golden('bary_cells.npz')
    data = core.Data(g['in_ti'], g['in_we'], g['in_TSTART'], g['in_TSTOP'], g['in_LIVETIME'], g['in_exposure'],
                     timeref=timeref, device=DEV)
    with pytest.raises(NotImplementedError):
        data.get_cells(tcell=3600., use_barycenter=True, quiet=True)
    data.set_barycenter_converters(*_bary_converters(g))
    cells = data.get_cells(tcell=3600., use_barycenter=True, quiet=True)
    assert np.array_equal(data._topo_edges, g[tag + '_topo_edges'])
    _same(_describe(cells), g, tag + '_', exact=('starts', 'stops', 'counts', 'exp'))
    ts = data.get_cells(tcell=1800., use_barycenter=True, time_series_only=True, trim_zero_exposure=False,
                        quiet=True)
    n = lambda x: x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    for k in ('starts', 'stops', 'exp', 'sexp', 'bexp', 'counts', 'alt_starts', 'alt_stops'):
        assert np.array_equal(n(getattr(ts, k)), g['%s_ts_%s' % (tag, k)]), k
    assert np.allclose(n(ts.weights), g[tag + '_ts_weights'], rtol=1e-12, atol=1e-13)
    assert np.allclose(n(ts.weights2), g[tag + '_ts_weights2'], rtol=1e-12, atol=1e-13)
    assert ts.timesys == str(g[tag + '_ts_timesys']) == 'barycenter'


@pytest.mark.parametrize('tag', ['all', 'part'])
def test_photon_cells_match_reference(golden, tag, capsys):
    """tcell=None: one cell per photon, boundaries at the exposure midpoints, cells
    broken at gaps of the exposure table."""
    g = golden('photon_cells.npz')
    data = _data(g)
    a, b = g[tag + '_range']
    cells = data.get_photon_cells(a, b)
    assert 'have %d photons' % len(g[tag + '_ti']) in capsys.readouterr().out
    n = lambda x: x.cpu().numpy()
    assert len(cells) == len(g[tag + '_starts'])
    assert np.array_equal(n(cells.hi) - n(cells.lo), np.ones(len(cells), dtype=np.int64))
    assert np.array_equal(n(cells.ti), g[tag + '_ti'])
    assert np.array_equal(n(cells.we), g[tag + '_we'])
    assert np.array_equal(n(cells.starts), g[tag + '_starts'])
    assert np.array_equal(n(cells.stops), g[tag + '_stops'])
    assert np.allclose(n(cells.sexp), g[tag + '_exp'], rtol=1e-12, atol=0)
    assert np.allclose(n(cells.sexp / cells.bexp), g[tag + '_SonB'], rtol=1e-12, atol=0)
    if tag == 'all':
        via = data.get_cells(tcell=None, use_barycenter=False, quiet=True)
        assert np.array_equal(n(via.starts), g['via_get_cells_starts'])
    with pytest.raises(ValueError):
        data.get_photon_cells(g['in_TSTART'][0] - 1.0, b)
    with pytest.raises(ValueError):
        data.get_photon_cells(a, g['in_TSTOP'][-1] + 1.0)
