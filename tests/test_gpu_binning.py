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


def _ragged_sources():
    """Sources that differ in everything a descriptor carries: photon count, the span of the
    exposure table, brightness; one of them with a single photon."""
    from godot_amd import core, synth
    out = []
    for k, (nph, years) in enumerate(((30_000, 0.30), (1, 0.05), (70_000, 0.50), (5_000, 0.11), (40_000, 0.50))):
        s = synth.synth(nph, years=years, seed=400 + k)
        d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=DEV)
        d.S, d.B = s['S'] * (1.0 + 0.2 * k), s['B']
        out.append((s, d))
    return out


@pytest.mark.parametrize('span', [(None, None), ('inner', 'inner')])
@pytest.mark.parametrize('tcell', [600., 86400., 3e7])
def test_cells_of_many_sources_in_one_call_equal_the_per_source_calls(tcell, span):
    """godot_bin_sources writes, for every source, the bits that godot_exposure_at and
    godot_bin_photons give on that source's own edges: cell lengths from many cells a source
    to one cell that is longer than the table (core.py:2064-2068), the whole table and an
    inner span (the clipping of core.py:2040-2057 differs from source to source), with and
    without the trimming of empty cells; and the first source against the oracle."""
    from godot_amd import core
    srcs = _ragged_sources()
    datas = [d for _, d in srcs]
    t0 = max(d._t0 for d in datas) + 1234.5 if span[0] else None
    t1 = min(d._t1 for d in datas[2:]) - 777.25 if span[1] else None    # past the end of the short tables
    for trim in (True, False):
        one = core.get_cells_batch(datas, tstart=t0, tstop=t1, tcell=tcell, trim_zero_exposure=trim)
        each = core.get_cells_batch(datas, tstart=t0, tstop=t1, tcell=tcell, trim_zero_exposure=trim,
                                    per_source_calls=True)
        assert np.array_equal(one.cell_offsets, each.cell_offsets)
        assert torch.equal(one.source, each.source)
        for nm in ('lo', 'hi', 'starts', 'stops', 'sexp', 'bexp', 'we', 'ti'):
            a, b = getattr(one.cells, nm), getattr(each.cells, nm)
            assert a.shape == b.shape and torch.equal(a.view(torch.int64), b.view(torch.int64)), nm
    if span[0] is None:
        s = srcs[0][0]
        d = datas[0]
        b = orc.bin_cells(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['exposure'], s['LIVETIME'], d.S, d.B, d.E,
                          tcell=tcell)
        one = core.get_cells_batch(datas, tcell=tcell)
        c0, c1 = int(one.cell_offsets[0]), int(one.cell_offsets[1])
        assert c1 - c0 == len(b['counts'])
        assert np.array_equal(_n(one.cells.hi - one.cells.lo)[c0:c1], b['counts'])
        assert_same(_n(one.cells.starts)[c0:c1], b['starts'], "b['starts']")
        assert_same(_n(one.cells.sexp)[c0:c1], b['sexp'], "b['sexp']")


def test_batch_trims_on_exposure_as_the_single_source_call_does():
    """The cells a batch keeps are those with exposure > 0 (core.py:2110-2122), not those with
    source exposure > 0: ten-minute cells over orbital gaps, where the two running-sum
    differences round apart, against Data.get_cells(trim_zero_exposure=True) source by source
    and against the oracle's mask; a source whose weights sum to zero keeps its cells."""
    from godot_amd import core
    srcs = _ragged_sources()
    datas = [d for _, d in srcs]
    datas[3].S = 0.0                       # no source photons expected: sexp = 0 in every cell
    one = core.get_cells_batch(datas, tcell=600., trim_zero_exposure=True)
    for k, ((s, d)) in enumerate(srcs):
        c0, c1 = int(one.cell_offsets[k]), int(one.cell_offsets[k + 1])
        own = d.get_cells(tcell=600., use_barycenter=False, trim_zero_exposure=True, quiet=True)
        assert c1 - c0 == len(own) and len(own) > 0
        for nm in ('starts', 'stops', 'sexp', 'bexp'):
            a, b = getattr(one.cells, nm)[c0:c1], getattr(own, nm)
            assert torch.equal(a.view(torch.int64), b.view(torch.int64)), (k, nm)
        assert torch.equal((one.cells.hi - one.cells.lo)[c0:c1], own.hi - own.lo)
        b = orc.bin_cells(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['exposure'], s['LIVETIME'], d.S, d.B, d.E,
                          tcell=600.)
        assert_same(_n(one.cells.starts)[c0:c1], b['starts'], 'starts of the cells the oracle keeps')
        assert np.array_equal(_n(one.cells.hi - one.cells.lo)[c0:c1], b['counts'])


def test_bin_sources_with_dead_time_and_rejected_descriptors():
    """The dead time of every cell equals Data.get_exposure(edges, cumulative=False)'s; a
    descriptor table whose cells overlap, a non-positive cell length and a short workspace
    are refused before anything is launched."""
    import ctypes
    from godot_amd import ops, _lib
    srcs = _ragged_sources()[:3]
    desc = []
    for _, d in srcs:
        t0, t1, tc, nc = d._uniform_cells(None, None, 7200., True)
        desc.append(dict(tstart=d.TSTART, tstop=d.TSTOP, exposure=d.exposure, livetime=d.LIVETIME,
                         table=d._plain_exposure_table(), ti=d.ti, s_over_e=d.S / d.E, b_over_e=d.B / d.E,
                         t0=t0, tcell=tc, ncell=nc))
    starts, stops, exp, sexp, bexp, dead, lo, hi, source = ops.bin_sources(desc, deadtime_too=True)
    c = p = 0
    for k, (_, d) in enumerate(srcs):
        n = desc[k]['ncell']
        edges = torch.arange(n + 1, dtype=torch.float64, device=DEV) * desc[k]['tcell'] + desc[k]['t0']
        ref = d.get_exposure(edges, deadtime_too=True, cumulative=False)
        for got, want in zip((exp, sexp, bexp, dead), ref):
            assert torch.equal(got[c:c + n].view(torch.int64), want.view(torch.int64))
        assert torch.equal(starts[c:c + n], edges[:-1]) and torch.equal(stops[c:c + n], edges[1:])
        rl, rh, _ = ops.bin_photons(d.ti, desc[k]['t0'], edges[1:].contiguous())
        assert torch.equal(lo[c:c + n], rl + p) and torch.equal(hi[c:c + n], rh + p)
        assert bool((source[c:c + n] == k).all())
        c += n
        p += d.ti.numel()
    # the C ABI refuses what its kernels could not index
    rec = np.zeros(2, dtype=ops.SOURCE_CELLS)
    for k in range(2):
        d = srcs[k][1]
        for nm, t in (('tstart', d.TSTART), ('tstop', d.TSTOP), ('exposure', d.exposure), ('livetime', d.LIVETIME),
                      ('table', d._plain_exposure_table()), ('ti', d.ti)):
            rec[nm][k] = t.data_ptr()
        rec['n_ft2'][k], rec['nph'][k] = d.TSTART.numel(), d.ti.numel()
        rec['t0'][k], rec['tcell'][k], rec['ncell'][k] = d._t0, 7200., 10
    rec['cell_offset'][:] = [0, 10]
    out = [torch.empty(20, dtype=torch.float64, device=DEV) for _ in range(3)]
    idx = [torch.empty(20, dtype=torch.int64, device=DEV) for _ in range(2)]
    nws = _lib.lib.godot_bin_sources_workspace_bytes(2, 20)
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)

    def call(r, total=20, wbytes=nws):
        return _lib.lib.godot_bin_sources(r.ctypes.data, 2, total, 0, 0, out[0].data_ptr(), out[1].data_ptr(),
                                          out[2].data_ptr(), 0, idx[0].data_ptr(), idx[1].data_ptr(), 0,
                                          ws.data_ptr(), wbytes, 0)
    assert call(rec) == 0
    torch.cuda.synchronize()
    bad = rec.copy()
    bad['cell_offset'][1] = 5                      # overlaps the cells of source 0
    assert call(bad) != 0 and 'overlap' in _lib.last_error()
    assert call(rec, total=19) != 0                # the last source passes total_cells
    bad = rec.copy()
    bad['tcell'][0] = 0.0
    assert call(bad) != 0 and 'cell length' in _lib.last_error()
    bad = rec.copy()
    bad['tcell'][1] = np.inf
    assert call(bad) != 0 and 'cell length' in _lib.last_error()
    bad = rec.copy()
    bad['t0'][0] = np.nan
    assert call(bad) != 0 and 'first edge' in _lib.last_error()
    bad = rec.copy()
    bad['table'][1] = 0
    assert call(bad) != 0 and 'null pointer' in _lib.last_error()
    assert call(rec, wbytes=nws - 1) != 0 and 'workspace' in _lib.last_error()
    torch.cuda.synchronize()


def test_packed_sources_are_binned_without_copying_their_photons():
    """core.pack_sources moves the photon lists into one allocation; the batch then hands out
    a view of it (no copy of 16 bytes a photon per call) and every array of the batch equals
    the unpacked batch's; a source whose list is replaced afterwards falls back to the copy."""
    from godot_amd import core
    datas = [d for _, d in _ragged_sources()]
    ref = core.get_cells_batch(datas, tcell=3600.)
    before = [(d.ti.clone(), d.we.clone()) for d in datas]
    assert core.pack_sources(datas) is datas
    for d, (ti, we) in zip(datas, before):
        assert torch.equal(d.ti, ti) and torch.equal(d.we, we)
    got = core.get_cells_batch(datas, tcell=3600.)
    assert got.cells.we.data_ptr() == datas[0].we.data_ptr()
    assert got.cells.ti.data_ptr() == datas[0].ti.data_ptr()
    assert np.array_equal(got.cell_offsets, ref.cell_offsets) and torch.equal(got.source, ref.source)
    for nm in ('lo', 'hi', 'starts', 'stops', 'sexp', 'bexp', 'we', 'ti'):
        a, b = getattr(got.cells, nm), getattr(ref.cells, nm)
        assert a.shape == b.shape and torch.equal(a.view(torch.int64), b.view(torch.int64)), nm
    # single sources still work on their views
    one = datas[2].get_cells(tcell=3600., use_barycenter=False, quiet=True)
    c0, c1 = int(ref.cell_offsets[2]), int(ref.cell_offsets[3])
    assert torch.equal(one.sexp, ref.cells.sexp[c0:c1]) and torch.equal(one.hi - one.lo, (ref.cells.hi - ref.cells.lo)[c0:c1])
    datas[1].we = datas[1].we.clone()
    again = core.get_cells_batch(datas, tcell=3600.)
    assert again.cells.we.data_ptr() != datas[0].we.data_ptr()
    assert torch.equal(again.cells.we, ref.cells.we) and torch.equal(again.cells.lo, ref.cells.lo)


def test_bin_sources_beyond_one_grid_of_sources():
    """More sources than a launch grid has rows (65 535): the call makes a second pair of
    launches for the rest, and every source -- here 66 000 descriptors of two photon lists in
    turn -- gets the cells of its own list, its photon indices shifted by what lies before it."""
    from godot_amd import ops
    pair = [d for _, d in _ragged_sources()[:3:2]]
    desc = []
    for d in pair:
        t0, t1, tc, nc = d._uniform_cells(None, d._t0 + 5 * 86400. + 1.0, 86400., True)
        assert nc == 5
        desc.append(dict(tstart=d.TSTART, tstop=d.TSTOP, exposure=d.exposure, livetime=d.LIVETIME,
                         table=d._plain_exposure_table(), ti=d.ti, s_over_e=d.S / d.E, b_over_e=d.B / d.E,
                         t0=t0, tcell=tc, ncell=nc))
    ref = ops.bin_sources(desc)
    nsrc = 66_000
    out = ops.bin_sources([desc[k & 1] for k in range(nsrc)])
    n0, n1 = pair[0].ti.numel(), pair[1].ti.numel()
    k = torch.arange(nsrc, device=DEV)
    before = (k + 1) // 2 * n0 + k // 2 * n1                       # photons ahead of source k
    for i in (0, 1, 2, 3, 4):                                      # starts, stops, exp, sexp, bexp
        got = out[i].view(nsrc // 2, 10)
        assert torch.equal(got.view(torch.int64), ref[i].view(torch.int64).expand(nsrc // 2, 10))
    for i in (6, 7):                                               # lo, hi
        own = torch.cat([ref[i][:5], ref[i][5:] - n0])             # indices into each list alone
        got = out[i].view(nsrc, 5) - before[:, None]
        assert torch.equal(got.view(nsrc // 2, 10), own.expand(nsrc // 2, 10))
    assert torch.equal(out[8].view(nsrc, 5), k[:, None].expand(nsrc, 5))


@pytest.mark.parametrize('trim', [True, False])
def test_time_series_of_many_sources_equal_the_per_source_calls(trim):
    """core.get_timeseries_batch gives every source the CellTimeSeries of its own
    get_cells(..., time_series_only=True): edges, exposures, dead time and counts bit for bit,
    the weight moments to the last bits (their sums may be formed in another order when the
    cells of all sources share a launch), and the power spectrum of a source follows."""
    from godot_amd import core
    datas = [d for _, d in _ragged_sources()]
    t1 = min(d._t1 for d in datas[2:]) - 777.25
    got = core.get_timeseries_batch(datas, tstop=t1, tcell=300., trim_zero_exposure=trim)
    assert len(got) == len(datas)
    for d, ts in zip(datas, got):
        ref = d.get_cells(tstop=t1, tcell=300., time_series_only=True, use_barycenter=False,
                          trim_zero_exposure=trim, quiet=True)
        assert ts.timesys == ref.timesys and ts.minimum_exposure == ref.minimum_exposure
        for nm in ('starts', 'stops', 'exp', 'sexp', 'bexp', 'deadtime', 'counts'):
            a, b = getattr(ts, nm), getattr(ref, nm)
            assert a.shape == b.shape and torch.equal(a.view(torch.int64), b.view(torch.int64)), nm
        for nm in ('weights', 'weights2'):
            a, b = _n(getattr(ts, nm)), _n(getattr(ref, nm))
            assert a.shape == b.shape and np.all(np.abs(a - b) <= 1e-13 * np.maximum(np.abs(b), 1.0)), nm
    f, p0, p1, pb = core.power_spectrum_fft(got[2])
    ref = datas[2].get_cells(tstop=t1, tcell=300., time_series_only=True, use_barycenter=False,
                             trim_zero_exposure=trim, quiet=True)
    rf, r0, r1, rb = core.power_spectrum_fft(ref)
    assert np.array_equal(f, rf)
    for a, b in ((p0, r0), (p1, r1), (pb, rb)):
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin)
        assert np.all(np.abs(a[fin] - b[fin]) <= 1e-9 * np.maximum(np.abs(b[fin]), 1e-3 * np.abs(b[fin]).max()))


def test_batch_with_a_span_that_ends_before_a_source_begins():
    """A requested span that lies past the end of one source's table leaves that source a cell
    of no length (core.py:2064-2068 with tstop clipped below tstart): the batch then takes the
    per-source calls, whatever they make of it, instead of refusing the descriptor."""
    from godot_amd import core
    datas = [d for _, d in _ragged_sources()]
    t0 = datas[1]._t1 + 5000.0                     # beyond the 0.05-year table of source 1
    outcome = []
    for per in (True, False):
        try:
            b = core.get_cells_batch(datas, tstart=t0, tcell=86400., trim_zero_exposure=False, per_source_calls=per)
            outcome.append((None, b))
        except Exception as e:                     # the same failure either way, if it is one
            outcome.append((type(e), None))
    assert outcome[0][0] is outcome[1][0]
    if outcome[0][0] is None:
        a, b = outcome[0][1], outcome[1][1]
        assert np.array_equal(a.cell_offsets, b.cell_offsets)
        for nm in ('lo', 'hi', 'starts', 'stops', 'sexp', 'bexp'):
            assert torch.equal(getattr(a.cells, nm).view(torch.int64), getattr(b.cells, nm).view(torch.int64)), nm
    ts = core.get_timeseries_batch(datas, tstart=t0, tcell=86400., trim_zero_exposure=False)
    assert len(ts) == len(datas)


def test_bin_sources_with_rescaled_exposure_rows():
    """A descriptor may carry per-row source and background exposures (what apply_scalers
    leaves in Data.sexposure / bexposure, core.py:2282-2306) with the table built from them;
    mixed with a plain source in one batch, each gets the bits of its own per-source call."""
    from godot_amd import ops
    (s0, d0), (s1, d1) = _ragged_sources()[:3:2]
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    n = d0.exposure.numel()
    sexposure = (d0.exposure * (d0.S / d0.E) * (0.5 + torch.rand(n, dtype=torch.float64, device=DEV, generator=g))).contiguous()
    bexposure = (d0.exposure * (d0.B / d0.E) * (0.5 + torch.rand(n, dtype=torch.float64, device=DEV, generator=g))).contiguous()
    table = ops.exposure_table_rows(d0.TSTART, d0.TSTOP, d0.exposure, sexposure, bexposure, d0.LIVETIME)
    desc = []
    for d, extra in ((d0, dict(sexposure=sexposure, bexposure=bexposure, table=table)),
                     (d1, dict(table=d1._plain_exposure_table()))):
        t0, t1, tc, nc = d._uniform_cells(None, None, 5400., True)
        desc.append(dict(tstart=d.TSTART, tstop=d.TSTOP, exposure=d.exposure, livetime=d.LIVETIME, ti=d.ti,
                         s_over_e=d.S / d.E, b_over_e=d.B / d.E, t0=t0, tcell=tc, ncell=nc, **extra))
    out = ops.bin_sources(desc, deadtime_too=True)
    n0, n1 = desc[0]['ncell'], desc[1]['ncell']
    e0 = torch.arange(n0 + 1, dtype=torch.float64, device=DEV) * desc[0]['tcell'] + desc[0]['t0']
    ref0 = ops.exposure_at_rows(d0.TSTART, d0.TSTOP, d0.exposure, sexposure, bexposure, d0.LIVETIME, table, e0,
                                cumulative=False, deadtime_too=True)
    e1 = torch.arange(n1 + 1, dtype=torch.float64, device=DEV) * desc[1]['tcell'] + desc[1]['t0']
    ref1 = d1.get_exposure(e1, deadtime_too=True, cumulative=False)
    for got, a, b in zip((out[2], out[3], out[4], out[5]), ref0, ref1):
        assert torch.equal(got[:n0].view(torch.int64), a.view(torch.int64))
        assert torch.equal(got[n0:].view(torch.int64), b.view(torch.int64))
    assert not torch.equal(out[3][:n0], out[2][:n0] * (d0.S / d0.E))      # the rescaled rows were used
    with pytest.raises(ValueError):
        ops.bin_sources([dict(desc[0], bexposure=None)])


def test_exposure_tables_built_on_several_streams_are_the_same_tables():
    """core.build_exposure_tables issues the table builds of many sources over several streams;
    every table equals, byte for byte in the part queries read, the one built alone, and the
    cells binned from them are the same."""
    from godot_amd import core, ops
    datas = [d for _, d in _ragged_sources()]
    alone = []
    for d in datas:
        t = ops.exposure_table(d.TSTART, d.TSTOP, d.exposure, d.LIVETIME, d.S / d.E, d.B / d.E)
        alone.append(t.clone())
    assert all(d._exposure_table is None for d in datas)
    assert core.build_exposure_tables(datas, nstream=3) is datas
    torch.cuda.synchronize()
    for d, t in zip(datas, alone):
        n = d.TSTART.numel()
        used = 80 * n                      # 32 bytes of running sums and 48 bytes of row a table row
        assert d._exposure_table is not None
        got = d._exposure_table
        assert got.numel() == t.numel()
        sums = -(-32 * n // 256) * 256     # the rows follow the sums on a 256-byte boundary
        assert torch.equal(got[:32 * n], t[:32 * n])
        assert torch.equal(got[sums:sums + 48 * n], t[sums:sums + 48 * n])
        assert used <= got.numel()
    core.build_exposure_tables(datas)      # nothing left to do
    a = core.get_cells_batch(datas, tcell=3600.)
    fresh = [d for _, d in _ragged_sources()]
    b = core.get_cells_batch(fresh, tcell=3600.)
    for nm in ('lo', 'hi', 'starts', 'stops', 'sexp', 'bexp'):
        assert torch.equal(getattr(a.cells, nm).view(torch.int64), getattr(b.cells, nm).view(torch.int64)), nm


def _checkpoint_case(case, n, rng):
    if case == 'realistic':
        return 27 * (3000 + 2000 * np.sin(np.arange(n) * 0.033) ** 2), np.full(n, 3.0)
    if case == 'ties':
        x1 = (2 * rng.integers(0, 1 << 30, n) + 1) * 2.0 ** -14 + 2.0 ** 22
        x4 = (2 * rng.integers(0, 8, n) + 1) * 2.0 ** -30
        x4[0] = 2.0 ** 23
        return x1, x4
    if case == 'negzero':
        # sums that begin with -0.0 keep its sign until the first +0.0 or nonzero row
        x1 = np.where(rng.random(n) < 0.5, -0.0, 0.0)
        x1[:3] = -0.0
        x1[n // 2:] = np.where(rng.random(n - n // 2) < 0.3, 1.5, -0.0)
        x4 = np.full(n, -0.0)
        x4[n - 1] = 0.0
        return x1, x4
    if case == 'zeros':
        x1 = np.where(rng.random(n) < 0.7, 0.0, rng.uniform(0, 1e5, n))
        x1[:min(5000, n // 2)] = 0.0
        return x1, np.zeros(n)
    return 10.0 ** rng.uniform(-12, 9, n), 2.0 ** rng.integers(-40, 30, n).astype(np.float64)


def test_checkpoints_of_many_tables_are_numpy_cumsum_at_every_sixteenth_row():
    """godot_exposure_checkpoints builds, for many tables in one call, the running sums that
    godot_exposure_table builds for one, and keeps those of rows 15, 31, ... and of the last
    row: numpy.cumsum's bits, for tables whose lengths are and are not multiples of sixteen and
    of the chunk of the parallel sums, tables too short for the parallel sums among them, and
    for series that tie at every addition."""
    from godot_amd import ops
    rng = np.random.default_rng(77)
    tables, want = [], []
    for k, (case, n) in enumerate((('realistic', 150_001), ('ties', 150_000), ('zeros', 40_960), ('range', 8_192),
                                   ('realistic', 8_191), ('range', 17), ('ties', 16), ('zeros', 1),
                                   ('ties', 263_000), ('negzero', 40), ('negzero', 700), ('range', 512),
                                   ('ties', 513), ('realistic', 1_025), ('zeros', 4_097))):
        x1, x4 = _checkpoint_case(case, n, rng)
        se, be = 0.173 + 0.01 * k, 0.827 - 0.01 * k
        z = torch.zeros(n, dtype=torch.float64, device=DEV)
        tables.append(dict(tstart=z, tstop=torch.from_numpy(x4).to(DEV), exposure=torch.from_numpy(x1).to(DEV),
                           livetime=z, s_over_e=se, b_over_e=be))
        with np.errstate(all='ignore'):
            full = np.stack([np.cumsum(x1), np.cumsum(se * x1), np.cumsum(be * x1), np.cumsum((x4 - 0.0) - 0.0)], 1)
        rows = np.minimum(np.arange(15, n + 15, 16), n - 1)
        want.append(full[rows])
    got = ops.exposure_checkpoints(tables)
    torch.cuda.synchronize()
    assert len(got) == len(tables)
    for k, (g, w) in enumerate(zip(got, want)):
        g = g[:32 * len(w)].view(torch.float64).view(len(w), 4).cpu().numpy()
        for c in range(4):
            assert _same_bits(np.ascontiguousarray(g[:, c]), np.ascontiguousarray(w[:, c])), \
                (k, c, int(np.argmax(g[:, c] != w[:, c])))
    # the full tables of the short ones hold the same sums (signs of zero included)
    for k in (9, 10, 13):
        t = tables[k]
        n = t['tstart'].numel()
        full = ops.exposure_table(t['tstart'], t['tstop'], t['exposure'], t['livetime'], t['s_over_e'], t['b_over_e'])
        full = full[:32 * n].view(torch.float64).view(n, 4).cpu().numpy()
        rows = np.minimum(np.arange(15, n + 15, 16), n - 1)
        for c in range(4):
            assert _same_bits(np.ascontiguousarray(full[rows, c]), np.ascontiguousarray(want[k][:, c])), (k, c)
    # every table alone: a call of its own takes the same route as the table took among the others
    for k in (0, 3, 4, 9, 11):
        g = ops.exposure_checkpoints([tables[k]])[0]
        nb = 32 * len(want[k])
        assert torch.equal(g[:nb], got[k][:nb]), k
    assert ops.exposure_checkpoints([]) == []


def _ragged_and_short_sources():
    from godot_amd import core, synth
    out = _ragged_sources()
    s = synth.synth(900, years=0.004, seed=499)           # 3 321 rows: the serial sums
    d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=DEV)
    out.insert(2, (s, d))
    return out


@pytest.mark.parametrize('tcell', [600., 86400., 3e7])
def test_cells_read_from_checkpoints_equal_those_read_from_the_full_tables(tcell):
    """Cells of many sources from the compact running sums (every sixteenth row kept, the rows
    between added at the query from the left): the bits of the full tables, cell for cell, in
    every output of godot_bin_sources, for cells shorter and longer than a table row, and in
    the time series built from them."""
    from godot_amd import core
    full = [d for _, d in _ragged_and_short_sources()]
    compact = [d for _, d in _ragged_and_short_sources()]
    for trim in (True, False):
        a = core.get_cells_batch(full, tcell=tcell, trim_zero_exposure=trim, compact_tables=False)
        b = core.get_cells_batch(compact, tcell=tcell, trim_zero_exposure=trim, compact_tables=True)
        assert np.array_equal(a.cell_offsets, b.cell_offsets)
        assert torch.equal(a.source, b.source)
        for nm in ('lo', 'hi', 'starts', 'stops', 'sexp', 'bexp'):
            x, y = getattr(a.cells, nm), getattr(b.cells, nm)
            assert x.shape == y.shape and torch.equal(x.view(torch.int64), y.view(torch.int64)), nm
    assert all(d._checkpoints is not None and d._exposure_table is None for d in compact)
    assert all(d._checkpoints is None and d._exposure_table is not None for d in full)
    n = full[0].TSTART.numel()
    assert compact[0]._checkpoints.numel() <= 2 * n + 256 < 80 * n <= full[0]._exposure_table.numel()
    ta = core.get_timeseries_batch(full, tcell=tcell)
    tb = core.get_timeseries_batch(compact, tcell=tcell)           # picks the tables the sources have
    for x, y in zip(ta, tb):
        for nm in core.CellTimeSeries._KEYS:
            u, v = getattr(x, nm), getattr(y, nm)
            assert u.shape == v.shape and torch.equal(u.view(torch.int64), v.view(torch.int64)), nm
    # first source against the oracle
    s, d = _ragged_and_short_sources()[0]
    got = core.get_cells_batch(compact, tcell=tcell)
    b = orc.bin_cells(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['exposure'], s['LIVETIME'], d.S, d.B, d.E,
                      tcell=tcell)
    c0, c1 = int(got.cell_offsets[0]), int(got.cell_offsets[1])
    assert c1 - c0 == len(b['counts'])
    assert np.array_equal(_n(got.cells.hi - got.cells.lo)[c0:c1], b['counts'])
    assert_same(_n(got.cells.sexp)[c0:c1], b['sexp'], "b['sexp']")
    assert_same(_n(got.cells.bexp)[c0:c1], b['bexp'], "b['bexp']")


def test_checkpoints_of_rescaled_exposure_rows():
    """Checkpoints built from per-row source and background exposures give godot_bin_sources
    the bits of the godot_exposure_table_rows table."""
    from godot_amd import ops
    (s0, d0), (s1, d1) = _ragged_sources()[:3:2]
    g = torch.Generator(device=DEV)
    g.manual_seed(6)
    n = d0.exposure.numel()
    sexposure = (d0.exposure * (d0.S / d0.E) * (0.5 + torch.rand(n, dtype=torch.float64, device=DEV, generator=g))).contiguous()
    bexposure = (d0.exposure * (d0.B / d0.E) * (0.5 + torch.rand(n, dtype=torch.float64, device=DEV, generator=g))).contiguous()
    table = ops.exposure_table_rows(d0.TSTART, d0.TSTOP, d0.exposure, sexposure, bexposure, d0.LIVETIME)
    rows = [dict(tstart=d0.TSTART, tstop=d0.TSTOP, exposure=d0.exposure, livetime=d0.LIVETIME, sexposure=sexposure,
                 bexposure=bexposure),
            dict(tstart=d1.TSTART, tstop=d1.TSTOP, exposure=d1.exposure, livetime=d1.LIVETIME, s_over_e=d1.S / d1.E,
                 b_over_e=d1.B / d1.E)]
    cks = ops.exposure_checkpoints(rows)
    outs = []
    for tabs in ((dict(table=table), dict(table=d1._plain_exposure_table())),
                 (dict(checkpoints=cks[0]), dict(checkpoints=cks[1]))):
        desc = []
        for d, r, tab in zip((d0, d1), rows, tabs):
            t0, t1, tc, nc = d._uniform_cells(None, None, 5400., True)
            desc.append(dict(r, ti=d.ti, s_over_e=d.S / d.E, b_over_e=d.B / d.E, t0=t0, tcell=tc, ncell=nc, **tab))
        outs.append(ops.bin_sources(desc, deadtime_too=True))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int64), b.view(torch.int64))
    with pytest.raises(ValueError):
        ops.bin_sources([dict(desc[0], checkpoints=cks[0][:100])])
