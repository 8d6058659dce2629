"""world_size-2 gloo tests (CPU) of the sharding / result-collection layer.
The per-rank compute is a numpy stand-in (the oracle): the GPU kernels are
covered by the -m gpu tests, this covers the multi-process path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from godot_amd import dist as gd
        from oracle import godot_oracle as orc
        rng = np.random.default_rng(5)
        ncell = 37
        counts = rng.integers(0, 40, ncell)
        counts[5] = 0
        off = np.concatenate([[0], np.cumsum(counts)])
        we = np.clip(rng.beta(0.3, 1.5, off[-1]), 1e-6, 1 - 1e-6)
        S = np.maximum(counts * 0.17, 0.05)

        def fit_fn(c0, c1):
            out = np.empty((c1 - c0, 2))
            with np.errstate(all='ignore'):
                for i, c in enumerate(range(c0, c1)):
                    cl = orc.CellLike(we[off[c]:off[c + 1]], S[c], 0.2)
                    a = cl.get_max()
                    out[i] = a, cl.get_ts(a)
            return torch.from_numpy(out)

        rec = gd.fit_cells_sharded(counts, fit_fn)
        ref = fit_fn(0, ncell)
        ok1 = bool(torch.equal(rec, ref))
        c0, c1 = gd.cell_shard(counts)

        t = np.sort(rng.uniform(0, 1e6, 400))
        w = rng.uniform(0.1, 0.9, 400)
        nf = 101

        def search_fn(j0, n):
            return torch.from_numpy(orc.z2m(t, w, 1e-3, 2e-7, n, m=2, j0=j0))

        zmax, jmax, spec = gd.search_sharded(nf, search_fn, gather_spectrum=True)
        full = orc.z2m(t, w, 1e-3, 2e-7, nf, m=2)
        ok2 = (jmax == int(np.argmax(full))) and (zmax == full.max()) and \
            np.array_equal(spec.numpy(), full)
        tv, ti_ = gd.search_topk_sharded(nf, search_fn, k=7)
        want = np.lexsort((np.arange(nf), -full))[:7]
        ok2 = ok2 and np.array_equal(ti_, want) and np.array_equal(tv, full[want])
        # the same search with the photons split over the ranks: sums are all-reduced
        p0, p1 = gd.photon_shard(len(t))

        def sums_fn(j0, n):
            C, Sn = orc.z2m_trig_sums(t[p0:p1], w[p0:p1], 1e-3, 2e-7, j0 + np.arange(n), 2)
            return (torch.from_numpy(np.stack([C.T, Sn.T], axis=2).copy()),
                    torch.tensor([np.sum(w[p0:p1] ** 2)]))

        def finish_fn(sums, sumw2):
            return (2. / sumw2) * (sums ** 2).sum(dim=(1, 2))

        zps = gd.search_photon_sharded(nf, sums_fn, finish_fn, tile=40).numpy()
        ok2 = ok2 and len(zps) == nf and np.abs(zps - full).max() <= 1e-12 * full.max()
        # Bayesian-Blocks fitness rows, sharded by row
        nb, npt = 23, 16
        lo = rng.uniform(0, 2, nb)
        dom = np.linspace(lo, lo + rng.uniform(0.5, 3, nb), npt).T.copy()
        cod = -((dom - dom.mean(axis=1, keepdims=True)) * 2) ** 2

        def rows_fn(r0, r1):
            with np.errstate(all='ignore'):
                rows = [orc.bb_fitness(dom, cod, 0, i + 1) for i in range(r0, r1)]
            cat = lambda k: np.concatenate([r[k] for r in rows]) if rows else np.empty(0)
            return torch.from_numpy(cat(0)), torch.from_numpy(cat(1))

        fv, fa = gd.bb_fitness_sharded(nb, rows_fn)
        rv, ra = rows_fn(0, nb)
        ok3 = bool(torch.equal(fv, rv)) and bool(torch.equal(fa, ra))
        # FFT power spectrum with its five transforms dealt over the ranks (numpy's rfft as
        # the per-rank transform): every rank ends with the oracle's spectra, to the bit
        nbin = 300
        cnt = rng.poisson(3.0, nbin).astype(float)
        W = cnt * rng.uniform(0.1, 0.3, nbin)
        WW = W * rng.uniform(0.2, 0.6, nbin)
        Sx = np.full(nbin, W.mean() * 1.1)
        Bx = np.full(nbin, (cnt - W).mean() * 0.9)
        starts = 100. * np.arange(nbin)
        ref_ps = orc.power_spectrum_fft(starts, starts + 100., cnt, W, WW, Sx, Bx)
        l = 4 * (len(ref_ps[0]) - 1)
        Wb, WbWb = cnt - W, cnt - 2 * W + WW
        series = [W - Sx, Wb - Bx, WW, WbWb, W - WW]
        made = []

        def series_fn(i):
            made.append(i)
            return series[i]

        def rfft_fn(x, n):
            f = np.fft.rfft(x, n=n)
            return torch.from_numpy(np.stack([f.real, f.imag], axis=1))

        def estimate_fn(*parts):
            c = [p_.numpy()[:, 0] + 1j * p_.numpy()[:, 1] for p_ in parts]
            return orc.power_spectrum_from_parts(*c)

        ps = gd.power_spectrum_fft_sharded(series_fn, l, rfft_fn, estimate_fn)
        ok4 = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ps, ref_ps[1:]))
        ok4 = ok4 and made == gd.transform_shard(5) and made == list(range(rank, 5, world))
        # four-step transform: the exchange puts every rank's columns side by side into whole rows.
        # Stand-ins for the passes: "columns" hands out slices of known arrays [8193][cols][2],
        # "rows" checks that it received the right rows of all columns and marks its bins
        l4 = 1 << 23
        shapes_all = [(8193, 256)] * 3 + [(8193, 512)] * 2
        full = [torch.arange(r * c * 2, dtype=torch.float64).reshape(r, c, 2) + 1000.0 * i
                for i, (r, c) in enumerate(shapes_all)]

        def columns_fn(part, nparts):
            return None, [f[:, part * (f.shape[1] // nparts):(part + 1) * (f.shape[1] // nparts)].contiguous()
                          for f in full]

        def shapes_fn(nrows):
            return [(nrows, c) for _, c in shapes_all]

        seen = {}

        def rows_fn(v_rows, dc, m0, m1):
            at, ok = 0, True
            for f in full:
                n_ = (m1 - m0) * f.shape[1] * 2
                ok = ok and bool(torch.equal(v_rows[at:at + n_].view(m1 - m0, f.shape[1], 2), f[m0:m1]))
                at += n_
            seen['ok'], seen['rows'] = ok and at == v_rows.numel(), (m0, m1)
            out = torch.zeros(3, 16)
            out[:, rank] = 1.0 + rank
            return out[0], out[1], out[2]

        z = torch.zeros(10, dtype=torch.float64)
        p0_, p1_, pb_ = gd.power_spectrum_fft_four_step(z, z, z, z, z, z, l4, columns_fn, rows_fn, shapes_fn)
        want = torch.zeros(16)
        want[0], want[1] = 1.0, 2.0
        ok4 = ok4 and seen['ok'] and seen['rows'] == gd.four_step_rows(world, rank) and bool(torch.equal(p1_, want))
        # per-source records of a job dealt round-robin come back in source order on every rank
        mine = gd.source_shard(7)
        recs = gd.gather_source_records(7, torch.tensor([[float(i), 10.0 * i] for i in mine]).reshape(-1, 2))
        ok4 = ok4 and bool(torch.equal(recs, torch.tensor([[float(i), 10.0 * i] for i in range(7)])))
        q.put((rank, ok1, ok2 and ok3 and ok4, (c0, c1), gd.source_shard(7)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_fit_and_search():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] and r[2] for r in res)
    (a0, a1), (b0, b1) = res[0][3], res[1][3]
    assert a0 == 0 and a1 == b0 and b1 == 37 and 0 < a1 < 37
    assert res[0][4] == [0, 2, 4, 6] and res[1][4] == [1, 3, 5]


def test_shard_helpers_single_process():
    sys.path.insert(0, REPO)
    from godot_amd import dist as gd
    counts = np.array([100, 0, 0, 1, 1, 1, 50, 50, 0, 3])
    cover = []
    for r in range(4):
        c0, c1 = gd.cell_shard(counts, r, 4)
        cover += list(range(c0, c1))
    assert cover == list(range(len(counts)))
    tot = 0
    for r in range(3):
        j0, n = gd.frequency_shard(10, r, 3)
        assert j0 == tot
        tot += n
    assert tot == 10
    assert gd.frequency_shard(2, 3, 4) == (2, 0)
    # five transforms over up to five ranks; further ranks get none
    assert [gd.transform_shard(5, r, 2) for r in range(2)] == [[0, 2, 4], [1, 3]]
    assert [gd.transform_shard(5, r, 8) for r in range(8)] == [[0], [1], [2], [3], [4], [], [], []]
    assert gd.transform_shard(5, 0, 1) == [0, 1, 2, 3, 4]
    b = gd.split_by_weight(np.array([0, 10, 20, 30, 40]), 2)
    assert list(b) == [0, 2, 4]
    # fitness rows: contiguous cover, work (sum of row lengths) within a row of equal
    edges = [gd.bb_row_shard(1000, r, 4) for r in range(4)]
    assert edges[0][0] == 0 and edges[-1][1] == 1000
    assert all(edges[r][1] == edges[r + 1][0] for r in range(3))
    work = [sum(range(a + 1, b + 1)) for a, b in edges]
    assert max(work) - min(work) <= 2 * 1000
