"""Sharding of the hot path over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).

The path shards without any data-path exchange:
  * cell fits / profile grids: contiguous cell ranges balanced by photon
    count; every rank reads only its slice of the photon array;
  * Z^2_m searches: contiguous ranges of trial-frequency indices; the photon
    list is replicated (16 MB for 1e6 photons);
  * multi-source batches: sources dealt round-robin;
  * Bayesian-Blocks fitness rows: contiguous row ranges balanced by row length
    (row i costs i + 1 cell profiles); the profiles are replicated;
  * the FFT power spectrum: its five real transforms dealt over up to five
    ranks, the parts of the spectra that the estimators read all-gathered.
The only collectives there are result collection: an all-gather of
fixed-width per-cell records (padded to the largest shard) and, for searches,
an all-reduce(max) / all-gather of per-rank maxima.  Both are latency-bound
(KBs to a few MB), one hop on the xGMI mesh.

One variant has a real exchange step: a Z^2_m search whose photon list is
split over the ranks (a list too long to replicate, or already spread by the
loader).  Every rank forms the trig sums of its photons for a tile of
frequencies, the tiles are added with all-reduce(sum) (2 m doubles per
frequency), and every rank finishes the statistic from the totals.
"""
import numpy as np
import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def split_by_weight(cum, nparts):
    """Boundaries b[0..nparts] of contiguous ranges of items with about equal
    total weight; cum[i] = total weight of items [0, i) (len n+1, cum[0] = 0)."""
    cum = np.asarray(cum)
    n = len(cum) - 1
    targets = cum[-1] * (np.arange(1, nparts) / float(nparts))
    cuts = np.searchsorted(cum, targets, side='left')
    b = np.concatenate([[0], np.clip(cuts, 0, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


def cell_shard(counts, rank=None, world=None, group=None):
    """(first_cell, last_cell_exclusive) of this rank: contiguous cells,
    balanced by photon count (+1 per cell so that empty cells also spread)."""
    r, w = _world(group)
    rank = r if rank is None else rank
    world = w if world is None else world
    c = np.asarray(counts, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(c + 1)])
    b = split_by_weight(cum, world)
    return int(b[rank]), int(b[rank + 1])


def frequency_shard(nf, rank=None, world=None, group=None):
    """(first_index, count) of this rank's contiguous slice of nf trial frequencies."""
    r, w = _world(group)
    rank = r if rank is None else rank
    world = w if world is None else world
    per = -(-nf // world)
    j0 = min(rank * per, nf)
    return j0, min(per, nf - j0)


def source_shard(nsource, rank=None, world=None, group=None):
    """Indices of the sources this rank owns (round-robin)."""
    r, w = _world(group)
    rank = r if rank is None else rank
    world = w if world is None else world
    return list(range(rank, nsource, world))


def gather_source_records(nsource, rec, group=None):
    """Per-source records of a job whose sources were dealt round-robin (source_shard): `rec`
    holds one row per source of this rank, in the order of source_shard(nsource); returns the
    (nsource, ...) array of all sources in source order, on every rank.  One all-gather."""
    rank, world = _world(group)
    sizes = [len(range(r, nsource, world)) for r in range(world)]
    if rec.shape[0] != sizes[rank]:
        raise ValueError('%d records for the %d sources of rank %d' % (rec.shape[0], sizes[rank], rank))
    if world == 1:
        return rec
    allrec = all_gather_ragged(rec, sizes=sizes, group=group)
    order = torch.tensor([i for r in range(world) for i in range(r, nsource, world)], dtype=torch.int64,
                         device=allrec.device)
    out = torch.empty_like(allrec)
    out[order] = allrec
    return out


def run_sources_sharded(nsource, make_sources, tcell_fit, tcell_ps, group=None, **kw):
    """The multi-source job (core.run_sources) over the ranks of a process group: the sources
    are dealt round-robin, every rank builds its own (`make_sources(indices)` -> list of
    core.Data on this rank's GPU), runs them as one job, and the per-source summaries (the
    largest P_0, P_1, P_b and where they stand, the number of fitted cells and the sum of their
    TS) are all-gathered: no data-path collective, a few dozen bytes a source at the end.
    Returns (summaries[nsource, 8], this rank's SourceResults)."""
    from . import core
    mine = source_shard(nsource, group=group)
    dev = None
    if mine:
        res = core.run_sources(make_sources(mine), tcell_fit, tcell_ps, **kw)
        dev = res.alpha.device
        ts_sum = torch.stack([x.sum() for x in res.batch.split(res.ts)]).cpu().numpy()
        ncl = np.diff(res.batch.cell_offsets).astype(np.float64)
        rec = np.concatenate([res.peak_power, res.peak_freq, ncl[:, None], ts_sum[:, None]], axis=1)
    else:
        res, rec = None, np.empty((0, 8))
    if dev is None:
        dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    out = gather_source_records(nsource, torch.from_numpy(rec).to(dev), group=group)
    return out.cpu().numpy(), res


def all_gather_ragged(local, sizes=None, group=None):
    """Concatenate per-rank tensors with different first dimensions, in rank
    order, on every rank.  One all-gather of the sizes, one of the padded
    records.  The result lives where `local` does (under the gloo backend the records
    travel through the host and come back)."""
    rank, world = _world(group)
    if world == 1:
        return local
    home = local.device
    local = local.to(_device_of(group))
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    if sizes is None:
        allsz = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(allsz, n, group=group)
        sizes = [int(x) for x in allsz.cpu()]
    nmax = max(sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    out = out.view((world, nmax) + tuple(local.shape[1:]))
    return torch.cat([out[r, :sizes[r]] for r in range(world)], dim=0).to(home)


def fit_cells_sharded(counts, fit_fn, group=None):
    """Fit this rank's cells and return the records of ALL cells on every rank.

    counts : host array of photons per cell (same on every rank)
    fit_fn : callable(first_cell, last_cell) -> (n_local, k) tensor of per-cell
             records (e.g. alpha, TS, flags) computed on this rank's device; it
             should fit ops.CellRanges.shard(first_cell, last_cell, photons=counts[first_cell:
             last_cell].sum()): the shard keeps the whole weight array and tells the kernels
             how many photons its own cells hold (they size their shape for that)
    """
    c0, c1 = cell_shard(counts, group=group)
    rec = fit_fn(c0, c1)
    if rec.shape[0] != c1 - c0:
        raise ValueError('fit_fn returned %d records for %d cells' % (rec.shape[0], c1 - c0))
    rank, world = _world(group)
    sizes = [cell_shard(counts, r, world)[1] - cell_shard(counts, r, world)[0] for r in range(world)]
    return all_gather_ragged(rec, sizes=sizes, group=group)


def search_topk_sharded(nf, search_fn, k=10, group=None):
    """The k largest values of the statistic over nf trial frequencies and their
    global frequency indices, on every rank: each rank searches its slice
    (search_fn(j0, n) -> (n,) tensor), keeps its own k best, and one all-gather of
    2 k doubles per rank is reduced to the global k best.  Returns (values,
    indices) as host arrays, largest first; ties go to the lower index."""
    rank, world = _world(group)
    j0, n = frequency_shard(nf, group=group)
    z = search_fn(j0, n)
    kk = min(k, n)
    rec = torch.full((k, 2), -np.inf, dtype=torch.float64, device=z.device)
    rec[:, 1] = -1.0
    if kk > 0:
        v, i = torch.topk(z, kk)
        rec[:kk, 0] = v.to(torch.float64)
        rec[:kk, 1] = (i + j0).to(torch.float64)
    if world > 1:
        rec = rec.to(_device_of(group))
        allrec = torch.empty((world * k, 2), dtype=torch.float64, device=rec.device)
        dist.all_gather_into_tensor(allrec, rec.contiguous(), group=group)
        rec = allrec
    r = rec.cpu().numpy()
    r = r[r[:, 1] >= 0]
    order = np.lexsort((r[:, 1], -r[:, 0]))[:k]
    return r[order, 0], r[order, 1].astype(np.int64)


def search_sharded(nf, search_fn, gather_spectrum=False, group=None):
    """Run search_fn(j0, n) -> (n,) statistic tensor on this rank's slice of nf
    trial frequencies.  Returns (global_max, global_argmax) and, if
    gather_spectrum, the full spectrum on every rank."""
    rank, world = _world(group)
    j0, n = frequency_shard(nf, group=group)
    z = search_fn(j0, n)
    if n > 0:
        zmax, imax = torch.max(z, dim=0)
        best = torch.stack([zmax.to(torch.float64), (imax + j0).to(torch.float64)])
    else:
        best = torch.tensor([-np.inf, -1.0], dtype=torch.float64, device=z.device)
    if world > 1:
        best = best.to(_device_of(group))
        allbest = torch.empty((world, 2), dtype=torch.float64, device=best.device)
        dist.all_gather_into_tensor(allbest, best.unsqueeze(0).contiguous(), group=group)
        k = int(torch.argmax(allbest[:, 0]).cpu())
        best = allbest[k]
    spec = None
    if gather_spectrum:
        sizes = [frequency_shard(nf, r, world)[1] for r in range(world)]
        spec = all_gather_ragged(z, sizes=sizes, group=group)
    return float(best[0].cpu()), int(best[1].cpu()), spec


def photon_shard(nph, rank=None, world=None, group=None):
    """(first_photon, last_photon_exclusive) of this rank's contiguous share of nph photons."""
    r, w = _world(group)
    rank = r if rank is None else rank
    world = w if world is None else world
    return (nph * rank) // world, (nph * (rank + 1)) // world


def search_photon_sharded(nf, sums_fn, finish_fn, tile=1 << 20, group=None):
    """Z^2_m / H over nf trial frequencies with the photons split over the ranks.

    sums_fn   : callable(j0, n) -> (sums (n, m, 2) tensor, sumw2 (1,) tensor) for
                THIS rank's photons (ops.z2m_sums with the rank's photon slice)
    finish_fn : callable(total_sums, total_sumw2) -> (n,) statistic tensor
                (ops.z2m_finish)
    Frequencies are processed in tiles of `tile`: one all-reduce(sum) of
    2 m tile doubles each (32 MB at m = 2, tile = 2**20), so the exchange of one
    tile can overlap the kernel of the next on another stream.  Returns the full
    spectrum on every rank.
    """
    rank, world = _world(group)
    out = []
    w2_total = None
    for j0 in range(0, nf, tile):
        n = min(tile, nf - j0)
        sums, sumw2 = sums_fn(j0, n)
        if world > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
            if w2_total is None:
                w2_total = sumw2.clone()
                dist.all_reduce(w2_total, op=dist.ReduceOp.SUM, group=group)
        else:
            w2_total = sumw2
        out.append(finish_fn(sums, w2_total))
    if not out:
        return torch.empty(0, dtype=torch.float64)
    return torch.cat(out)


def bb_row_shard(ncell, rank=None, world=None, group=None):
    """(first_row, last_row_exclusive) of this rank's share of the fitness rows:
    row i sums i + 1 profiles, so the ranges are balanced by sum(i + 1)."""
    r, w = _world(group)
    rank = r if rank is None else rank
    world = w if world is None else world
    i = np.arange(ncell + 1, dtype=np.int64)
    b = split_by_weight(i * (i + 1) // 2, world)
    return int(b[rank]), int(b[rank + 1])


def bb_fitness_sharded(ncell, rows_fn, group=None):
    """Compute this rank's fitness rows and return the packed lower triangle of
    ALL rows on every rank.

    rows_fn : callable(first_row, last_row) -> (fit_val, fit_arg) tensors holding
              those rows packed one after the other (ops.bb_fitness(..., rows=...))
    """
    rank, world = _world(group)
    r0, r1 = bb_row_shard(ncell, group=group)
    fv, fa = rows_fn(r0, r1)
    tri = lambda r: r * (r + 1) // 2
    if fv.numel() != tri(r1) - tri(r0):
        raise ValueError('rows_fn returned %d values for rows %d..%d' % (fv.numel(), r0, r1))
    sizes = [tri(bb_row_shard(ncell, r, world)[1]) - tri(bb_row_shard(ncell, r, world)[0])
             for r in range(world)]
    rec = all_gather_ragged(torch.stack([fv, fa], dim=1), sizes=sizes, group=group)
    return rec[:, 0].contiguous(), rec[:, 1].contiguous()



def transform_shard(ntransform=5, rank=None, world=None, group=None):
    """Indices of the transforms (of ntransform) that this rank computes: dealt round-robin
    over the first min(world, ntransform) ranks, the others get none."""
    r, w = _world(group)
    rank = r if rank is None else rank
    world = w if world is None else world
    return list(range(rank, ntransform, min(world, ntransform))) if rank < ntransform else []


def power_spectrum_fft_sharded(series_fn, l, rfft_fn, estimate_fn, group=None, device=None):
    """power_spectrum_fft (reference core.py:2683-2741) with its five real transforms of
    length l dealt over the ranks (one whole transform per rank and turn: the spectrum of a
    twelve-year light curve in 100-s bins, l = 2**25, is 268 MB, a single GPU transforms it
    in a fraction of a millisecond, so a transform is not split further).

    series_fn   : callable(i) -> the i-th series on this rank's device, i = 0 .. 4 for
                  W - S, Wb - B, WW, WbWb, W - WW (unpadded; formed only for the
                  transforms this rank owns)
    rfft_fn     : callable(series, l) -> (l/2 + 1, 2) tensor of (re, im), numpy.fft.rfft(x, n=l)
    estimate_fn : callable(f1, f2, g3, g4, g5) -> the estimators from the five parts,
                  each (nk, 2) with nk = l/4 + 1 (ops.power_spectrum_from_parts)
    Every rank cuts its transforms to the nk entries the estimators read (the first nk bins
    of the first two, the even bins of the other three: half of each spectrum), one
    all-gather collects them (5 * 16 * nk bytes in all; 671 MB at l = 2**25), and every rank
    forms the estimators.  Returns what estimate_fn returns."""
    rank, world = _world(group)
    nk = l // 4 + 1
    mine = transform_shard(5, rank, world)
    parts = {}
    for i in mine:
        f = rfft_fn(series_fn(i), l)
        parts[i] = (f[:nk] if i < 2 else f[0:2 * nk:2][:nk]).contiguous()
    if world > 1:
        slots = -(-5 // min(world, 5))
        ref = next(iter(parts.values())) if parts else None
        cdev = _device_of(group)                       # where the collective runs (the host under gloo)
        # where the estimators are formed: the caller's device, that of this rank's own transforms, or
        # (a rank that owns none, i.e. more than five ranks) the current GPU if there is one
        if device is not None:
            dev = torch.device(device)
        elif ref is not None:
            dev = ref.device
        elif torch.cuda.is_available():
            dev = torch.device('cuda', torch.cuda.current_device())
        else:
            dev = cdev
        dtype = ref.dtype if ref is not None else torch.float64
        buf = torch.zeros((slots, nk, 2), dtype=dtype, device=cdev)
        for s_, i in enumerate(mine):
            buf[s_] = parts[i].to(cdev)
        allbuf = torch.empty((world * slots, nk, 2), dtype=dtype, device=cdev)
        dist.all_gather_into_tensor(allbuf, buf, group=group)
        allbuf = allbuf.view(world, slots, nk, 2)
        for r in range(min(world, 5)):
            for s_, i in enumerate(transform_shard(5, r, world)):
                if i not in parts:
                    parts[i] = allbuf[r, s_].to(dev)
    return estimate_fn(parts[0], parts[1], parts[2], parts[3], parts[4])


def four_step_rows(world, rank=None):
    """Row ranges of the four-step transform: the 8 192 + 1 stored rows in `world` contiguous
    ranges (the last takes the lone row 8 192).  Returns the list of edges, or (first, end) of
    `rank`."""
    half = 8192
    per = half // world
    edges = [r * per for r in range(world)] + [half + 1]
    return edges if rank is None else (edges[rank], edges[rank + 1])


def power_spectrum_fft_four_step(W, WW, S, Wb, WbWb, B, l, columns_fn, rows_fn, shapes_fn, group=None,
                                 gather=True):
    """ONE power spectrum (core.py:2598-2741) whose transforms are split over the ranks: every
    rank holds the whole series (they are small: 8 bytes a bin), transforms its share of the
    COLUMNS of the five series laid out as 16 384 rows (columns_fn), the ranks exchange what
    they made so that each holds whole ROWS (an all-to-all: 28 l bytes over all ranks, sent
    point to point), and every rank transforms its rows and forms the estimators of the bins
    they hold (rows_fn).  With gather=True the three spectra are summed over the ranks
    (every bin is written by one rank, the others hold zeros there), else every rank returns
    its own bins.  The work and the bytes a rank moves through its memory go down with the
    number of ranks, which the replicated variant (power_spectrum_fft_sharded: whole transforms
    dealt out) cannot offer beyond five ranks and pays for with the full transforms gathered.

    columns_fn(part, nparts) -> (buffer, [five (8193, columns, 2) tensors])   ops.power_spectrum_columns
    rows_fn(v_rows, dc, m_first, m_end) -> (p0, p1, pb)                        ops.power_spectrum_rows
    shapes_fn(nrows) -> five (rows, columns) shapes of whole rows              ops.power_spectrum_part_shapes
    The number of ranks must divide the columns as godot_power_spectrum_parts_supported says
    (a power of two up to 8 for l = 2^23, up to 32 for l = 2^25)."""
    rank, world = _world(group)
    edges = four_step_rows(world)
    m_first, m_end = edges[rank], edges[rank + 1]
    _, parts = columns_fn(rank, world)
    home = parts[0].device
    dc = torch.stack([WW.sum(), WbWb.sum(), (W - WW).sum()])
    shapes = shapes_fn(m_end - m_first)
    v_rows = torch.empty(sum(2 * r * c for r, c in shapes), dtype=parts[0].dtype, device=home)
    views, at = [], 0
    for r, c in shapes:
        views.append(v_rows[at:at + 2 * r * c].view(r, c, 2))
        at += 2 * r * c
    if world == 1:
        for v, p in zip(views, parts):
            v.copy_(p[m_first:m_end])
    else:
        cdev = _device_of(group)
        peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
        todo, pieces = [], []
        for i, p in enumerate(parts):
            pc = p.to(cdev)
            got = []
            for s_ in range(world):
                if s_ == rank:
                    got.append(pc[m_first:m_end])
                    continue
                t = torch.empty((m_end - m_first,) + tuple(pc.shape[1:]), dtype=pc.dtype, device=cdev)
                got.append(t)
                todo.append(dist.P2POp(dist.irecv, t, peer(s_), group, tag=i))
                todo.append(dist.P2POp(dist.isend, pc[edges[s_]:edges[s_ + 1]], peer(s_), group, tag=i))
            pieces.append(got)
        for req in dist.batch_isend_irecv(todo):
            req.wait()
        for v, got in zip(views, pieces):
            # rank s made columns s * columns / world ...: side by side they are the whole rows
            v.copy_(torch.cat(got, dim=1).to(home))
    p0, p1, pb = rows_fn(v_rows, dc, m_first, m_end)
    if gather and world > 1:
        both = torch.stack([p0, p1, pb]).to(_device_of(group))
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        p0, p1, pb = (x.to(home) for x in both)
    return p0, p1, pb


def _device_of(group=None):
    """Device of the collectives of this process: the current GPU under nccl, the host under gloo."""
    if dist.is_initialized() and dist.get_backend(group) == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')
