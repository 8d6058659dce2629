"""GPU parity of exposure lookup and photon->cell binning (through the
reference-shaped Data.get_cells API) against the reference's golden vectors
and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import godot_oracle as orc
from conftest import assert_same

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _data(g):
    from godot_amd import core
    return core.Data(g['in_ti'], g['in_we'], g['in_TSTART'], g['in_TSTOP'], g['in_LIVETIME'],
                     g['in_exposure'])


def _n(t):
    return t.cpu().numpy()


def test_exposure_matches_reference(golden):
    g = golden('exposure.npz')
    d = _data(g)
    assert d.S == float(g['in_S']) and d.E == float(g['in_E'])
    c = d.get_exposure(g['times'], deadtime_too=True, cumulative=True)
    # the running sums are formed in numpy's order: bit-identical, not just close
    for k in range(4):
        assert_same(_n(c[k]), g['cumulative'][k], "g['cumulative'][k]")
    dd = d.get_exposure(g['times'], deadtime_too=True, cumulative=False)
    for k in range(4):
        assert_same(_n(dd[k]), g['differential'][k], "g['differential'][k]")


def test_exposure_between_fewer_than_two_times_is_empty(golden):
    """cumulative=False differences neighbouring query times (core.py:1836-1841): one time,
    or none, gives empty arrays like numpy's x[1:] - x[:-1], not an error."""
    g = golden('exposure.npz')
    d = _data(g)
    for times in (g['times'][:1], g['times'][:0]):
        for dead in (False, True):
            out = d.get_exposure(times, deadtime_too=dead, cumulative=False)
            ref = orc.exposure_at(g['in_TSTART'], g['in_TSTOP'], g['in_exposure'], g['in_LIVETIME'], d.S, d.B,
                                  d.E, times, deadtime_too=dead, cumulative=False)
            for a, b in zip(out, ref):
                assert (a is None and b is None) or (_n(a).shape == b.shape == (0,))
    one = d.get_exposure(g['times'][:1], deadtime_too=True, cumulative=True)
    for k in range(4):
        assert_same(_n(one[k]), g['cumulative'][k][:1], 'first cumulative value')


@pytest.mark.parametrize('name', ['cells_dense.npz', 'cells_sparse.npz'])
def test_get_cells_matches_reference(golden, name):
    g = golden(name)
    d = _data(g)
    cells = d.get_cells(tcell=float(g['tcell']), use_barycenter=False)
    assert len(cells) == len(g['counts'])
    assert_same(_n(cells.starts), g['starts'], "g['starts']")
    assert_same(_n(cells.stops), g['stops'], "g['stops']")
    lo, hi = _n(cells.lo), _n(cells.hi)
    assert np.array_equal(hi - lo, g['counts'])            # bit-exact integer part
    we = np.concatenate([g['in_we'][a:b] for a, b in zip(lo, hi)])
    ti = np.concatenate([g['in_ti'][a:b] for a, b in zip(lo, hi)])
    assert_same(we, g['we'], "g['we']")
    assert_same(ti, g['ti'], "g['ti']")
    assert_same(_n(cells.sexp), g['exp'], "g['exp']")
    assert_same(_n(cells.sexp) / _n(cells.bexp), g['SonB'], "g['SonB']")
    # a Cell materialised from the batch looks like the reference's
    c = cells[3]
    assert c.tstart == g['starts'][3] and len(c.we) == g['counts'][3]
    c.sanity_check()


def test_time_series_matches_reference(golden):
    g = golden('timeseries_fft.npz')
    d = _data(g)
    ts = d.get_cells(tcell=float(g['tcell']), time_series_only=True, trim_zero_exposure=False,
                     use_barycenter=False)
    assert_same(_n(ts.starts), g['starts'], "g['starts']")
    assert np.array_equal(_n(ts.counts), g['counts'])      # bit-exact
    for k, gk in (('exp', 'exp'), ('sexp', 'sexp'), ('bexp', 'bexp'), ('deadtime', 'deadtime')):
        assert_same(_n(getattr(ts, k)), g[gk], 'g[gk]')
    assert np.allclose(_n(ts.weights), g['weights'], rtol=1e-12, atol=1e-13)
    assert np.allclose(_n(ts.weights2), g['weights2'], rtol=1e-12, atol=1e-13)
    assert ts.tsamp() == g['stops'][0] - g['starts'][0]
    assert ts.ps_nbin() == len(g['starts']) // 4 + 1


def test_binning_edge_semantics():
    """Photon on an edge goes to the earlier cell; the first edge is closed,
    the last open (numpy searchsorted side='left' on stops)."""
    from godot_amd import ops
    dev = torch.device('cuda:0')
    stops = np.array([10., 20., 30.])
    ti = np.array([-1., 0., 0., 5., 10., 10., 10.000001, 20., 29.999, 30., 31.])
    start0 = 0.
    lo, hi, cnt = ops.bin_photons(torch.from_numpy(ti).to(dev), start0, torch.from_numpy(stops).to(dev))
    sl = slice(*np.searchsorted(ti, [start0, stops[-1]]))
    ref = np.bincount(np.searchsorted(stops, ti[sl]), minlength=3)
    assert np.array_equal(_n(cnt), ref)
    assert _n(lo)[0] == 1 and _n(hi)[-1] == 9
    # empty photon list, and a single cell
    lo, hi, cnt = ops.bin_photons(torch.empty(0, dtype=torch.float64, device=dev), 0.,
                                  torch.from_numpy(stops).to(dev))
    assert _n(cnt).sum() == 0
    lo, hi, cnt = ops.bin_photons(torch.from_numpy(ti).to(dev), 0., torch.tensor([30.], dtype=torch.float64, device=dev))
    assert _n(cnt)[0] == 8


def test_large_binning_matches_oracle_and_conserves_photons():
    """1e6 photons into 10-minute cells: integer layout bit-exact vs the oracle."""
    from godot_amd import core, synth
    s = synth.synth(1000000, years=1.0, seed=11)
    d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'])
    cells = d.get_cells(tcell=600., use_barycenter=False)
    b = orc.bin_cells(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['exposure'], s['LIVETIME'],
                      s['S'], s['B'], s['E'], tcell=600.)
    lo, hi = _n(cells.lo), _n(cells.hi)
    assert_same(hi - lo, b['counts'], "b['counts']")
    assert_same(_n(cells.starts), b['starts'], "b['starts']")
    # kept photons are exactly the oracle's, in order
    idx = np.concatenate([np.arange(a, c) for a, c in zip(lo[:2000], hi[:2000])])
    nn = len(idx)
    assert np.array_equal(s['we'][idx], b['weights'][:nn])
    assert (hi - lo).sum() == len(b['weights'])
    assert_same(_n(cells.sexp), b['sexp'], "b['sexp']")
    assert_same(_n(cells.bexp), b['bexp'], "b['bexp']")
    ts = d.get_cells(tcell=600., use_barycenter=False, time_series_only=True, trim_zero_exposure=False)
    assert int(_n(ts.counts).sum()) == np.searchsorted(s['ti'], _n(ts.stops)[-1]) - np.searchsorted(s['ti'], _n(ts.starts)[0])


def _table_columns(x1, x4, s_over_e, b_over_e):
    """Running sums of the four series through godot_exposure_table: exposure x1,
    its two multiples, and x4 as the dead time (TSTART = LIVETIME = 0, TSTOP = x4)."""
    from godot_amd import ops
    n = len(x1)
    z = torch.zeros(n, dtype=torch.float64, device=DEV)
    table = ops.exposure_table(z, torch.from_numpy(x4).to(DEV), torch.from_numpy(x1).to(DEV), z.clone(),
                               s_over_e, b_over_e)
    torch.cuda.synchronize()
    return table[:32 * n].view(torch.float64).view(n, 4).cpu().numpy()


def _same_bits(a, b):
    return np.array_equal(a.view(np.int64), b.view(np.int64))


@pytest.mark.parametrize('case', ['realistic', 'ties', 'zeros', 'range', 'negative', 'short', 'nan'])
def test_running_sums_are_numpy_cumsum_bit_for_bit(case):
    """The parallel form of the running sums (integer scan inside a binade,
    parity-dependent ties) must give numpy.cumsum's left-to-right result exactly."""
    rng = np.random.default_rng(sum(map(ord, case)))
    n = 150_001
    se, be = 0.173, 0.827
    if case == 'realistic':
        x1 = 27 * (3000 + 2000 * np.sin(np.arange(n) * 0.033) ** 2)
        x4 = np.full(n, 3.0)
    elif case == 'ties':
        # x1: odd multiples of 2^-14 (+ 2^22): every addition is an exact tie once s is in
        # [2^39, 2^40), u = 2^-13, from row 131072 on; x4: odd multiples of 2^-30 onto
        # s = 2^23 (u = 2^-29): ties from the second row on, s never leaves the binade
        x1 = (2 * rng.integers(0, 1 << 30, n) + 1) * 2.0 ** -14 + 2.0 ** 22
        x4 = (2 * rng.integers(0, 8, n) + 1) * 2.0 ** -30
        x4[0] = 2.0 ** 23
        assert 2.0 ** 39 < np.sum(x1[:140000]) < np.sum(x1) < 2.0 ** 40
    elif case == 'zeros':
        x1 = np.where(rng.random(n) < 0.7, 0.0, rng.uniform(0, 1e5, n))
        x1[:5000] = 0.0
        x4 = np.zeros(n)
    elif case == 'range':
        x1 = 10.0 ** rng.uniform(-12, 9, n)
        x4 = 2.0 ** rng.integers(-40, 30, n).astype(np.float64)
    elif case == 'negative':
        x1 = rng.uniform(0, 1e5, n)
        x4 = rng.normal(0.0, 1.0, n)
        x4[::3] = np.abs(x4[::3])
    elif case == 'short':
        n = 5000
        x1 = rng.uniform(0, 1e5, n)
        x4 = rng.uniform(0, 3, n)
    else:
        n = 20_000
        x1 = rng.uniform(0, 1e5, n)
        x1[12345] = np.nan
        x4 = rng.uniform(0, 3, n)
        x4[777] = np.inf
    got = _table_columns(x1, x4, se, be)
    with np.errstate(all='ignore'):
        want = [np.cumsum(x1), np.cumsum(se * x1), np.cumsum(be * x1), np.cumsum((x4 - 0.0) - 0.0)]
    for c in range(4):
        assert _same_bits(got[:, c], want[c]), (case, c, int(np.argmax(got[:, c].view(np.int64) != want[c].view(np.int64))))


def test_gather_columns_is_the_per_column_gather():
    """The cell table trimmed in one launch: float64 and int64 columns, more columns than
    one launch takes, an empty selection."""
    import torch
    from godot_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu')
    g.manual_seed(5)
    n = 100_003
    cols = []
    for a in range(19):
        if a % 3 == 2:
            cols.append(torch.randint(-2**62, 2**62, (n,), generator=g, dtype=torch.int64).to(dev))
        else:
            c = torch.randn(n, generator=g, dtype=torch.float64)
            c[::977] = float('nan')                       # payload bits must survive
            cols.append(c.to(dev))
    idx = torch.sort(torch.randperm(n, generator=g)[:40_001])[0].to(dev)
    out = ops.gather_columns(cols, idx)
    assert len(out) == len(cols)
    for c, o in zip(cols, out):
        assert o.dtype == c.dtype
        assert torch.equal(o.view(torch.int64), c[idx].view(torch.int64))
    empty = ops.gather_columns(cols[:3], idx[:0])
    assert all(e.numel() == 0 for e in empty)
