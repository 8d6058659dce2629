#!/usr/bin/env python3
"""Benchmark of the weighted-photon likelihood hot path on MI355X.

One "step" = one batched maximum-likelihood fit (alpha-hat + TS, the
reference's get_max + get_ts) of every cell of one photon list of the workload
    C2: 1e7 synthetic photons, 10 yr, 10-minute cells (zero-exposure cells
        trimmed), cells built by the GPU binning path,
with all inputs resident in HBM.  Every rank owns NLIST = 4 such lists (seeds
11 + 4 rank + j) and the steps go round them, so that 293 MB of other inputs
pass between two fits of the same list: more than the 256 MiB Infinity Cache
holds, the weights come from HBM.  The fits of different lists do not depend
on one another, so the steps are dealt over NSTREAM = 3 HIP streams: the next
fit fills the compute units that the tail of the last one leaves idle (one
stream: 46 - 49 us a step; three: 38 us, less than one launch alone takes).
The roofline figures are those of single launches on one stream, as the HIP
events on that stream time them; roofline.overlapped restates the timed steps
in the same units.  Weak scaling: the path has no exchange between ranks, so
the timed steps hold no collective; value = cells fitted per second by all
ranks.

`python bench.py --gpus N` with N > 1 and no launcher around it starts the N
ranks itself (fresh child processes, before this process touches a GPU).

Also reported (rank 0, "extra"):
  * strong scaling of three sharded jobs, collectives inside the timed
    region: C2's cells of ONE list split over the ranks by photon count, the
    per-cell records all-gathered every step; C4 (1e6 photons x 1e8 trial
    frequencies, m = 2) with the frequencies split over the ranks and the global
    top ten gathered; one C3 power spectrum with its five transforms dealt over
    the ranks and the cut transforms all-gathered;
  * the same fit on one list kept in the cache, the 200-point profile grid,
    dense cells, power_spectrum_fft at the C3 size, binning, Bayesian Blocks;
  * CPU baselines: the numpy oracle (a restatement of the reference's Python
    path) on one core and on all cores of the box, for the cell fit, the Z^2_2
    sums and the FFT power spectrum.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
FP64_PEAK_TFLOPS = 78.6    # MI355X vector FP64 peak
NLIST = 4                  # photon lists a rank rotates through
NSTREAM = max(1, int(os.environ.get('GODOT_BENCH_STREAMS', '3')))   # streams the timed fits are dealt over
# hardware-counted FP64 work of the Z^2_2 kernel per (photon, frequency) evaluation:
# profiles/r3_z2m_flops.json (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 passes on the timed kernel)
Z2M_FLOP_FILE = os.path.join(REPO, 'profiles', 'r3_z2m_flops.json')
FIT_TRAFFIC_FILE = os.path.join(REPO, 'profiles', 'r3_cell_fit_traffic.json')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--nph', type=int, default=10_000_000)
    ap.add_argument('--years', type=float, default=10.0)
    ap.add_argument('--tcell', type=float, default=600.0)
    ap.add_argument('--nfreq', type=int, default=100_000_000, help='trial frequencies of the C4 search')
    ap.add_argument('--no-extra', action='store_true', help='skip the extra measurements')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baselines')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    return ap.parse_args()


# ---------------------------------------------------------------------------
# launching the ranks
# ---------------------------------------------------------------------------
def spawn_ranks(n):
    """Start n ranks of this script as child processes (this process has not touched a
    GPU and never will) and pass on rank 0's JSON line, which goes to the shared stdout."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    sys.exit(rc)


# ---------------------------------------------------------------------------
# CPU baselines (the oracle; run before the process initialises the GPU)
# ---------------------------------------------------------------------------
def _cpu_sample(s, tcell):
    """Arrays of a leading slice (40 days) of the workload, small enough to hand to worker processes."""
    t_end = s['TSTART'][0] + 40 * 86400.
    m = s['ti'] < t_end
    r = s['TSTOP'] <= t_end
    return dict(ti=s['ti'][m], we=s['we'][m], TSTART=s['TSTART'][r], TSTOP=s['TSTOP'][r],
                exposure=s['exposure'][r], LIVETIME=s['LIVETIME'][r], S=s['S'], B=s['B'], E=s['E'],
                tcell=tcell)


def _cpu_fit_worker(p, seconds):
    """The oracle's per-cell get_max + get_ts over the sample, repeated for `seconds`: (cells, time, ncell)."""
    from oracle import godot_oracle as orc
    # binning itself is not timed
    b = orc.bin_cells(p['ti'], p['we'], p['TSTART'], p['TSTOP'], p['exposure'], p['LIVETIME'], p['S'], p['B'],
                      p['E'], tcell=p['tcell'])
    clls = orc.cells_from_bins(b)
    n = 0
    t0 = time.perf_counter()
    with np.errstate(all='ignore'):
        while True:
            for c in clls:
                a = c.get_max()
                c.get_ts(a)
                n += 1
            if time.perf_counter() - t0 > seconds:
                break
    return n, time.perf_counter() - t0, len(clls)


def _cpu_z2m_worker(p, seconds):
    """The oracle's Z^2_2 over blocks of 16 trial frequencies for `seconds`: (evaluations, time)."""
    from oracle import godot_oracle as orc
    n = 0
    j0 = p['j0']
    t0 = time.perf_counter()
    while True:
        orc.z2m(p['t'], p['w'], p['f0'], p['df'], 16, m=2, j0=j0)
        j0 += 16
        n += 16 * len(p['t'])
        if time.perf_counter() - t0 > seconds:
            break
    return n, time.perf_counter() - t0


def _cpu_fft_worker(p, _seconds):
    """One power_spectrum_fft of the oracle (five numpy rfft of length l and the estimators): (1, time)."""
    from oracle import godot_oracle as orc
    nb = p['nbins']
    rng = np.random.default_rng(13)
    cnt = rng.poisson(2.6, nb).astype(float)
    W, WW = cnt * 0.17, cnt * 0.08
    S = np.full(nb, 0.44)
    starts = 100. * np.arange(nb)
    t0 = time.perf_counter()
    out = orc.power_spectrum_fft(starts, starts + 100., cnt, W, WW, S, S * 5)
    return 1, time.perf_counter() - t0, 4 * (len(out[0]) - 1)


def _all_cores(fn, p, seconds, cores):
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(cores, mp_context=mp.get_context('spawn')) as ex:
        return list(ex.map(fn, [p] * cores, [seconds] * cores))


def cpu_baselines(s, args):
    """The oracle on the host cores: one core (the reference is single-threaded), then the
    same loop in one process per core.  Must run before this process touches the GPU: the
    workers are spawned."""
    from godot_amd import synth
    cores = max(1, min(16, os.cpu_count() or 1))
    sec = args.cpu_seconds
    p = _cpu_sample(s, args.tcell)
    n, dt, ncl = _cpu_fit_worker(p, sec)
    fit = dict(value=n / dt, unit='cells/s', cores=1, kind='port',
               sample='oracle get_max+get_ts over the first %d cells (40 days) of the same '
                      'workload, repeated for %.0f s on one core' % (ncl, dt))
    try:
        res = _all_cores(_cpu_fit_worker, p, sec / 2, cores)
        fit['all_cores'] = dict(value=sum(r[0] for r in res) / max(r[1] for r in res), unit='cells/s', cores=cores,
                                sample='the same loop in %d processes, %.0f s each' % (cores, sec / 2))
    except Exception as e:   # a box that cannot spawn workers still reports the one-core figure
        fit['all_cores'] = dict(value=None, error=repr(e))
    if args.no_extra:
        return fit, None
    other = {}
    # Z^2_2: the C4 photon list, a few blocks of trial frequencies
    zs = synth.synth(1_000_000, years=1.0, seed=17)
    zt = zs['ti'] - zs['ti'][0]
    zp = dict(t=zt, w=zs['we'], f0=1e-4, df=1.0 / (4 * float(zt[-1])), j0=0)
    n, dt = _cpu_z2m_worker(zp, sec / 2)
    z = dict(value=n / dt, unit='photon x frequency evaluations/s', cores=1, kind='port',
             sample='oracle z2m (m = 2) on the C4 photon list, blocks of 16 trial frequencies for %.0f s' % dt)
    try:
        res = _all_cores(_cpu_z2m_worker, zp, sec / 3, cores)
        z['all_cores'] = dict(value=sum(r[0] for r in res) / max(r[1] for r in res), cores=cores,
                              unit='photon x frequency evaluations/s')
    except Exception as e:
        z['all_cores'] = dict(value=None, error=repr(e))
    other['z2m'] = z
    # FFT power spectrum one size below C3 (l = 2^24: a 2^25 run takes twice as long and as much again in memory)
    fp = dict(nbins=1_893_456)
    _, dt, l = _cpu_fft_worker(fp, 0)
    f = dict(value=dt, unit='s per power_spectrum_fft', cores=1, kind='port', l=l, nbins=fp['nbins'],
             sample='oracle power_spectrum_fft (five numpy rfft of length 2^24 + estimators), once')
    try:
        res = _all_cores(_cpu_fft_worker, fp, 0, min(cores, 8))
        f['all_cores'] = dict(value=max(r[1] for r in res) / len(res), cores=min(cores, 8),
                              unit='s per power_spectrum_fft (one spectrum per process, all at once)')
    except Exception as e:
        f['all_cores'] = dict(value=None, error=repr(e))
    other['power_spectrum_fft'] = f
    return fit, other


# ---------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------
def run_rank(args):
    # Libraries write to stdout as well (RCCL prints a version banner when it starts):
    # file descriptor 1 points to stderr until the one JSON line is due.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    from godot_amd import synth

    # ---- workload: C2, NLIST photon lists per rank; CPU comparators first, while this
    # process has not initialised the GPU (they spawn workers)
    lists = [synth.synth(args.nph, years=args.years, seed=11 + NLIST * rank + j) for j in range(NLIST)]
    cpu = cpu_other = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu, cpu_other = cpu_baselines(lists[0], args)

    import torch
    dist = None
    ndev = torch.cuda.device_count()
    local_dev = local % max(ndev, 1)
    # GODOT_BENCH_BACKEND=gloo: a rehearsal of the multi-rank path on a box with fewer GPUs
    # than ranks (RCCL wants a GPU per rank); GODOT_BENCH_FORCE_DIST=1: the multi-rank code
    # with a single rank
    backend = os.environ.get('GODOT_BENCH_BACKEND', 'nccl')
    if world > 1 or os.environ.get('GODOT_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from godot_amd import core, ops
    from godot_amd import dist as gd

    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    cdev = dev if backend == 'nccl' else torch.device('cpu')     # where collectives take their tensors
    stream = torch.cuda.current_stream(dev)

    def event_time_ms(fn, iters):
        """Average device time of fn(i) over iters launches, HIP events on the launch stream."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i, (a, b) in enumerate(evs):
            a.record(stream)
            fn(i)
            b.record(stream)
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs]))

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].cpu())

    crs = []
    t_bin_first = t_bin_again = None
    data0 = None
    for j, s in enumerate(lists):
        data = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'], device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cells = data.get_cells(tcell=args.tcell, use_barycenter=False, quiet=True)
        torch.cuda.synchronize()
        if j == 0:
            t_bin_first = time.perf_counter() - t0
            # with the running sums of the exposure table cached: mean of five calls
            t0 = time.perf_counter()
            for _ in range(5):
                cells = data.get_cells(tcell=args.tcell, use_barycenter=False, quiet=True)
            torch.cuda.synchronize()
            t_bin_again = (time.perf_counter() - t0) / 5
            data0 = data
        crs.append(cells.ranges())
        del data, cells
    lists = lists[:1]
    torch.cuda.empty_cache()
    ncells = [cr.ncell for cr in crs]
    nphs = [int((cr.hi - cr.lo).sum().cpu()) for cr in crs]

    def step_one(i):
        return ops.cell_fit(crs[i % NLIST])

    # the timed steps: independent fits dealt over NSTREAM streams (outputs are allocated on
    # the stream that fills them; torch.cuda.synchronize() below waits for all of them)
    fit_streams = [torch.cuda.Stream(dev) for _ in range(NSTREAM)]

    def step(i):
        with torch.cuda.stream(fit_streams[i % NSTREAM]):
            return ops.cell_fit(crs[i % NLIST])

    torch.cuda.synchronize()
    # the same steps on one stream, for comparison (extra.single_stream)
    for i in range(min(args.warmup, 8)):
        step_one(i)
    torch.cuda.synchronize()
    nsingle = max(args.steps, 1000)          # 1000 fits: a region of about 50 ms
    t0 = time.perf_counter()
    for i in range(nsingle):
        out = step_one(i)
    torch.cuda.synchronize()
    single_s = time.perf_counter() - t0

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # One step is one fit (40 us): K steps alone would be a timed region of a millisecond or
    # so around asynchronous launches, and the figure would move by +-10 % with the jitter of
    # the launches.  Every step is therefore repeated `reps` times inside the timed region, so
    # that the region lasts at least 50 ms; ms_per_step stays the time of ONE fit.
    t0 = time.perf_counter()
    for i in range(4 * NLIST):
        step(i)
    torch.cuda.synchronize()
    est = max_over_ranks((time.perf_counter() - t0) / (4 * NLIST))
    reps = max(1, int(np.ceil(0.05 / (args.steps * est))))
    nfit = args.steps * reps
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(nfit):
        out = step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cells_done = float(sum(ncells[i % NLIST] for i in range(nfit)))

    if dist is not None:
        tot = torch.tensor([cells_done], dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        cells_done = float(tot[0].cpu())
    elapsed = max_over_ranks(elapsed)

    extra = dict(ncell_per_list=ncells, photons_in_cells_per_list=nphs, lists_per_rank=NLIST,
                 binning_first_call_s=t_bin_first, binning_cached_table_s=t_bin_again,
                 streams=NSTREAM,
                 single_stream=dict(ms_per_step=single_s / nsingle * 1e3, fits_timed=nsingle,
                                    cells_per_s_this_rank=float(sum(ncells[i % NLIST] for i in range(nsingle)))
                                    / single_s))

    # the four lists as ONE cell list fitted by one launch (390 MB touched per launch: more than
    # the last-level cache): what a fit costs when the ramp and the tail of a 46-us launch are
    # spread over four times the work
    if not args.no_extra:
        shift = np.concatenate([[0], np.cumsum([cr.nph for cr in crs])])
        joined = ops.CellRanges(torch.cat([cr.we for cr in crs]),
                                torch.cat([cr.lo + int(shift[j]) for j, cr in enumerate(crs)]),
                                torch.cat([cr.hi + int(shift[j]) for j, cr in enumerate(crs)]),
                                torch.cat([cr.S for cr in crs]), torch.cat([cr.B for cr in crs]),
                                photons=int(sum(nphs)))
        ops.cell_fit(joined)
        j_ms = event_time_ms(lambda i: ops.cell_fit(joined), 50)
        j_bytes = sum(8.0 * n + 8.0 * (c + 1) + 32.0 * c for n, c in zip(nphs, ncells))
        extra['four_lists_one_launch'] = dict(ms=j_ms, ms_per_list=j_ms / NLIST, cells=int(joined.ncell),
                                              algorithmic_GBps=j_bytes / (j_ms * 1e-3) / 1e9,
                                              frac_of_hbm_peak=j_bytes / (j_ms * 1e-3) / 8e12,
                                              cells_per_s=joined.ncell / (j_ms * 1e-3))
        del joined
        torch.cuda.empty_cache()

    # ---- strong scaling, collectives inside the timed region -----------------------
    # Every leg stands alone: one that fails (on every rank alike: the ranks run the same code on the
    # same sizes) leaves its error in `extra` and the others, and the headline above, are still reported.
    if not args.no_extra:
        def leg_c2():
            # (1) the cells of ONE photon list (seed 11 on every rank) split by photon count; every
            # step fits the rank's shard and all-gathers the (alpha, TS) records of all cells
            if rank == 0:
                cr_all = crs[0]
            else:
                s0 = synth.synth(args.nph, years=args.years, seed=11)
                d0 = core.Data(s0['ti'], s0['we'], s0['TSTART'], s0['TSTOP'], s0['LIVETIME'], s0['exposure'], device=dev)
                cr_all = d0.get_cells(tcell=args.tcell, use_barycenter=False, quiet=True).ranges()
                del s0, d0
            counts = (cr_all.hi - cr_all.lo).cpu().numpy()
            c0, c1 = gd.cell_shard(counts, rank, world)
            mine = cr_all.shard(c0, c1, photons=int(counts[c0:c1].sum()))
            sizes = [gd.cell_shard(counts, r, world)[1] - gd.cell_shard(counts, r, world)[0] for r in range(world)]

            def strong_step():
                a, t, _ = ops.cell_fit(mine)
                rec = torch.stack([a, t], dim=1)
                return gd.all_gather_ragged(rec.to(cdev), sizes=sizes)

            for _ in range(5):
                rec = strong_step()
            nstrong = max(10, args.steps // 4)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nstrong):
                rec = strong_step()
            torch.cuda.synchronize()
            barrier()
            t_strong = max_over_ranks(time.perf_counter() - t0) / nstrong
            extra['strong_c2_fit'] = dict(cells_per_s=len(counts) / t_strong, ms_per_step=t_strong * 1e3,
                                          ncell=int(len(counts)), n_gpus=world, records_gathered=int(rec.shape[0]),
                                          cells_on_rank0=int(c1 - c0),
                                          sharding='cells split by photon count (dist.cell_shard), (alpha, TS) '
                                                   'records all-gathered inside every timed step')
            del rec, mine

        def leg_c4():
            # (2) C4: 1e6 photons x nfreq trial frequencies, m = 2, frequencies split over the ranks,
            # global top ten gathered; the search runs in launches of 2^21 frequencies
            zs = synth.synth(1_000_000, years=1.0, seed=17)
            zt = torch.from_numpy(zs['ti'] - zs['ti'][0]).to(dev)
            zw = torch.from_numpy(zs['we']).to(dev)
            zspan = float(zs['ti'][-1] - zs['ti'][0])
            zf0, zdf = 1e-4, 1.0 / (4 * zspan)
            chunk = 1 << 21
            kernel_ms = []
            j0_mine, n_mine = gd.frequency_shard(args.nfreq, rank, world)
            # the workspace of a launch is not monotonic in its number of frequencies (fewer frequencies are
            # split over more groups of photons): sized for the full launches and for the last, shorter one
            zneed = max(ops._lib.lib.godot_z2m_workspace_bytes(zt.numel(), nn_, 2)
                        for nn_ in {chunk, n_mine % chunk or chunk, min(chunk, max(n_mine, 1))})
            zws = torch.empty(zneed, dtype=torch.uint8, device=dev)
            zbuf = torch.empty(n_mine, dtype=torch.float64, device=dev)

            def search_fn(j0, n):
                """The statistic on this rank's slice, launch by launch, into one buffer."""
                for k0 in range(0, n, chunk):
                    nn = min(chunk, n - k0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    zbuf[k0:k0 + nn] = ops.z2m(zt, zw, zf0, zdf, nn, m=2, j0=j0 + k0, workspace=zws)[0]
                    e1.record(stream)
                    kernel_ms.append((e0, e1, nn))
                return zbuf[:n]

            def topk_sharded():
                return gd.search_topk_sharded(args.nfreq, search_fn, k=10)

            ops.z2m(zt, zw, zf0, zdf, chunk, m=2, workspace=zws)        # warm-up launch
            torch.cuda.synchronize()
            kernel_ms.clear()
            barrier()
            t0 = time.perf_counter()
            top_v, top_i = topk_sharded()
            torch.cuda.synchronize()
            barrier()
            z_s = max_over_ranks(time.perf_counter() - t0)
            k_ms = sum(a.elapsed_time(b) for a, b, _ in kernel_ms)
            k_n = sum(nn for _, _, nn in kernel_ms)
            evals = float(zt.numel()) * args.nfreq / z_s
            zrec = dict(photon_freq_evals_per_s=evals, s=z_s, nph=zt.numel(), nfreq=args.nfreq, m=2, n_gpus=world,
                        z2_max=float(top_v[0]), argmax=int(top_i[0]),
                        top_ten_indices=[int(i) for i in top_i], top_ten_values=[float(v) for v in top_v],
                        kernel_evals_per_s_rank0=float(zt.numel()) * k_n / (k_ms * 1e-3),
                        sharding='frequencies split (dist.frequency_shard), photons replicated, per-rank top ten '
                                 'all-gathered')
            if os.path.exists(Z2M_FLOP_FILE):
                with open(Z2M_FLOP_FILE) as f:
                    zf = json.load(f)
                fpe = zf.get('fp64_flop_per_evaluation')
                if fpe:
                    zrec['hardware_counted_fp64_flop_per_eval'] = fpe
                    zrec['executed_frac_of_fp64_peak'] = fpe * evals / (FP64_PEAK_TFLOPS * 1e12 * world)
            extra['strong_c4_z2m'] = zrec
            del zws, zt, zw, zbuf
            torch.cuda.empty_cache()

        def leg_c3():
            # (3) C3: ONE power spectrum (3 786 912 bins of 100 s, l = 2^25) with its five transforms
            # dealt over the ranks; every step transforms the rank's series, all-gathers the five
            # transforms cut to what the estimators read and forms the estimators on every rank.
            # (On one GPU the whole call is a single fused pipeline: extra.power_spectrum_fft.)
            nb3, l3 = 3_786_912, 1 << 25
            gen3 = torch.Generator(device=dev)
            gen3.manual_seed(13)                                   # the same series on every rank
            cnt3 = torch.poisson(torch.full((nb3,), 2.6, dtype=torch.float64, device=dev), generator=gen3)
            W3, WW3 = cnt3 * 0.17, cnt3 * 0.08
            S3 = torch.full((nb3,), 0.44, dtype=torch.float64, device=dev)
            B3 = S3 * 5
            Wb3, WbWb3 = cnt3 - W3, cnt3 - 2 * W3 + WW3
            series3 = [W3 - S3, Wb3 - B3, WW3, WbWb3, W3 - WW3]

            # the four-step form where the ranks divide the columns (a power of two of them): every rank
            # transforms its columns, an all-to-all (point to point) gives every rank whole rows, the
            # estimators of a rank's rows are summed over the ranks; else whole transforms are dealt out
            four_step = world > 1 and ops.power_spectrum_parts_supported(nb3, l3, world)
            c3_method = ('four-step: columns split, all-to-all of 28 l bytes, rows split; every rank keeps the bins of its '
                         'rows (the largest P_1 is all-reduced)'
                         if four_step else ('one fused call' if world == 1 else
                                            'whole transforms dealt over the ranks (dist.transform_shard), cut transforms '
                                            'all-gathered, estimators on every rank'))

            def c3_step():
                if four_step:
                    return gd.power_spectrum_fft_four_step(
                        W3, WW3, S3, Wb3, WbWb3, B3, l3,
                        lambda part, nparts: ops.power_spectrum_columns(W3, WW3, S3, Wb3, WbWb3, B3, l3, part, nparts),
                        lambda v, dc, m0, m1: ops.power_spectrum_rows(v, dc, l3, m0, m1)[:3],
                        lambda nrows: ops.power_spectrum_part_shapes(l3, 1, nrows), gather=False)
                if world == 1:
                    return ops.power_spectrum_fft(W3, WW3, S3, Wb3, WbWb3, B3, l3)[:3]
                return gd.power_spectrum_fft_sharded(lambda i: series3[i], l3, ops.rfft,
                                                     lambda *parts: ops.power_spectrum_from_parts(*parts))

            for _ in range(2):
                c3_step()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                p3 = c3_step()
            torch.cuda.synchronize()
            barrier()
            t_c3 = max_over_ranks(time.perf_counter() - t0) / 5
            p1_max3 = max_over_ranks(float(torch.nan_to_num(p3[1][1:], nan=-np.inf).max().cpu()))
            extra['strong_c3_fft'] = dict(ms=t_c3 * 1e3, nbins=nb3, l=l3, n_gpus=world,
                                          exchanged_MB=(16.0 * (l3 + 3 * l3 // 4) * (world - 1) / world / 1e6
                                                        if four_step else (5 * 16 * (l3 // 4 + 1) / 1e6 if world > 1 else 0.0)),
                                          p0_at_bin_1=float(p3[0][1]), p1_at_bin_1000=float(p3[1][1000]), p1_max=p1_max3,
                                          sharding=c3_method)
            del cnt3, W3, WW3, S3, B3, Wb3, WbWb3, series3, p3
            torch.cuda.empty_cache()

        def leg_c5():
            # (4) C5: 256 sources x 1e6 photons dealt round-robin over the ranks; every rank runs its
            # sources as one job (core.run_sources, tables built inside) and the per-source summaries
            # are all-gathered.  The sources are made before the clock starts.
            nsrc5 = 256
            mine5 = gd.source_shard(nsrc5, rank, world)
            made5 = [core.Data(s_['ti'], s_['we'], s_['TSTART'], s_['TSTOP'], s_['LIVETIME'], s_['exposure'], device=dev)
                     for s_ in synth.synth_sources_device(nsrc5, dev, years=args.years, indices=mine5)]
            core.pack_sources(made5)

            def c5_step():
                for d_ in made5:
                    d_._checkpoints = None
                return gd.run_sources_sharded(nsrc5, lambda idx: made5, 86400., 300.)

            c5_step()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            summ5, _ = c5_step()
            torch.cuda.synchronize()
            barrier()
            t_c5 = max_over_ranks(time.perf_counter() - t0)
            extra['strong_c5_sources'] = dict(sources=nsrc5, sources_on_rank0=len(mine5), n_gpus=world, s=t_c5,
                                              sources_per_s=nsrc5 / t_c5,
                                              largest_p1_first_four=[float(x) for x in summ5[:4, 1]],
                                              sharding='sources dealt round-robin (dist.source_shard), every rank runs its '
                                                       'own as one job, 64-byte summaries all-gathered')
            del made5
            torch.cuda.empty_cache()

        for name_, leg_ in (('strong_c2_fit', leg_c2), ('strong_c4_z2m', leg_c4), ('strong_c3_fft', leg_c3),
                            ('strong_c5_sources', leg_c5)):
            try:
                leg_()
            except Exception as e:
                import traceback
                traceback.print_exc()
                extra[name_] = dict(error='%s: %s' % (type(e).__name__, e), n_gpus=world)
            torch.cuda.empty_cache()

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (cell_fit), HIP events on its stream -----
    nev = max(20, min(args.steps, 200))
    k_ms = event_time_ms(step_one, nev)
    alg = [8 * nphs[j] + 8 * (ncells[j] + 1) + 16 * ncells[j] + 16 * ncells[j] for j in range(NLIST)]
    alg_bytes = float(np.mean([alg[i % NLIST] for i in range(nev)]))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    # the same fit on one list every time: its 98 MB stay in the 256 MiB Infinity Cache
    r_ms = event_time_ms(lambda i: ops.cell_fit(crs[0]), nev)
    # HBM bytes per launch from the PMC passes (taken separately with rocprofv3 on this same
    # rotating workload and committed under profiles/; null when the workload differs)
    traffic = None
    kernel_name = 'cell_fit_group_b0_kernel<1, 40, true>'
    if os.path.exists(FIT_TRAFFIC_FILE) and args.nph == 10_000_000 and args.tcell == 600.0 and args.years == 10.0:
        with open(FIT_TRAFFIC_FILE) as f:
            tj = json.load(f)
        traffic = tj.get('traffic_bytes_per_launch')
        kernel_name = tj.get('kernel', kernel_name)
    roofline = dict(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic, kernel=kernel_name,
                    algorithmic_bytes_per_launch=alg_bytes, avg_launch_ms=k_ms,
                    inputs='%d photon lists in rotation (%.0f MB between two fits of one list)' % (
                        NLIST, sum(alg) / 1e6 - alg[0] / 1e6),
                    cache_resident=dict(avg_launch_ms=r_ms, achieved=alg[0] / (r_ms * 1e-3) / 1e9,
                                        frac=alg[0] / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        inputs='one list fitted over and over (98 MB: stays in the Infinity Cache)'),
                    # the timed steps themselves (launches of independent lists overlapping on
                    # NSTREAM streams): algorithmic bytes per step / time per step
                    overlapped=dict(streams=NSTREAM, ms_per_fit=elapsed / nfit * 1e3,
                                    achieved=alg_bytes / (elapsed / nfit) / 1e9,
                                    frac=alg_bytes / (elapsed / nfit) / 1e9 / HBM_PEAK_GBS))

    if not args.no_extra:
        cr = crs[0]
        ncell, nph_in = ncells[0], nphs[0]
        # 200-point profile grid on the same cells (FP64-ALU bound)
        ops.cell_logpdf(cr, dlogl=30, npt=200)
        p_ms = event_time_ms(lambda i: ops.cell_logpdf(cr, dlogl=30, npt=200), 5)
        extra['profile_grid'] = dict(cells_per_s=ncell / (p_ms * 1e-3), ms=p_ms,
                                     log_evals_per_s=200.0 * nph_in / (p_ms * 1e-3))
        torch.cuda.empty_cache()
        # the C2 light curve end to end through the reference-shaped classes, wall clock: profiles of all
        # cells (CellsLogLikelihood), then fluxes, errors and TS as host rows (get_raw_lightcurve)
        try:
            cells0 = data0.get_cells(tcell=args.tcell, use_barycenter=False, quiet=True)

            def wall_ms(fn, n=5):
                fn()
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in range(n):
                    r_ = fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0_) / n * 1e3, r_

            c_ms, cll0 = wall_ms(lambda: core.CellsLogLikelihood(cells0))
            l_ms, _ = wall_ms(cll0.get_raw_lightcurve)
            extra['light_curve'] = dict(ncell=len(cells0), profiles_ms=c_ms, raw_lightcurve_ms=l_ms,
                                        cells_per_s=len(cells0) / ((c_ms + l_ms) * 1e-3))
            del cells0, cll0
        except Exception as e:       # reported, never fatal to the headline
            extra['light_curve'] = dict(error=repr(e))
        torch.cuda.empty_cache()
        # dense cells (the per-source shape of C1 / C5): same photons in 2-hour cells
        dcr = data0.get_cells(tcell=7200., use_barycenter=False, quiet=True).ranges()
        ops.cell_fit(dcr)
        d_ms = event_time_ms(lambda i: ops.cell_fit(dcr), 20)
        dn = int((dcr.hi - dcr.lo).sum().cpu())
        extra['dense_cells'] = dict(ncell=dcr.ncell, photons_per_cell=dn / max(dcr.ncell, 1), ms=d_ms,
                                    cells_per_s=dcr.ncell / (d_ms * 1e-3),
                                    algorithmic_GBps=(8 * dn + 40 * dcr.ncell) / (d_ms * 1e-3) / 1e9)
        # power_spectrum_fft at the C3 size: 3 786 912 bins of 100 s -> l = 2^25
        nb = 3_786_912
        g = torch.Generator(device=dev)
        g.manual_seed(13)
        cnt = torch.poisson(torch.full((nb,), 2.6, dtype=torch.float64, device=dev), generator=g)
        W = cnt * 0.17
        WW = cnt * 0.08
        S = torch.full((nb,), 0.44, dtype=torch.float64, device=dev)
        B = S * 5
        l = 1 << 25
        Wb, WbWb = cnt - W, cnt - 2 * W + WW
        ops.power_spectrum_fft(W, WW, S, Wb, WbWb, B, l)
        f_ms = event_time_ms(lambda i: ops.power_spectrum_fft(W, WW, S, Wb, WbWb, B, l), 5)
        extra['power_spectrum_fft'] = dict(ms=f_ms, nbins=nb, l=l,
                                           algorithmic_GBps=5 * (8 * nb + 16 * (l // 4 + 1)) / (f_ms * 1e-3) / 1e9)
        del cnt, W, WW, S, B, Wb, WbWb
        torch.cuda.empty_cache()

        # the multi-source job (C5: 1000 sources x 1e6 photons over one spacecraft file, daily cells
        # fitted, 300-s series transformed) at 256 of its sources, through its driver: exposure
        # tables, binning, one fit launch, time series, a power spectrum per source.  The sources
        # are made on the device (a host-made source takes a quarter of a second).
        nsrc = 256
        srcs = [core.Data(s_['ti'], s_['we'], s_['TSTART'], s_['TSTOP'], s_['LIVETIME'], s_['exposure'], device=dev)
                for s_ in synth.synth_sources_device(nsrc, dev, years=args.years)]
        core.pack_sources(srcs)                    # photon lists in one allocation: batches are views
        core.run_sources(srcs[:8], 86400., 300., chunk=8)        # code objects load, the allocator grows
        for d_ in srcs:
            d_._checkpoints = None
        torch.cuda.synchronize()
        t0m = time.perf_counter()
        core.run_sources(srcs, 86400., 300., chunk=32)
        torch.cuda.synchronize()
        job_s = time.perf_counter() - t0m
        for d_ in srcs:
            d_._checkpoints = None
        stages = {}
        mres = core.run_sources(srcs, 86400., 300., chunk=32, timings=stages)
        mcr = mres.batch.cells.ranges()
        m_ms = event_time_ms(lambda i: ops.cell_fit(mcr), 10)
        extra['multi_source'] = dict(sources=nsrc, photons_per_source=1_000_000, table_rows=int(srcs[0].TSTART.numel()),
                                     cells_fitted=int(mcr.ncell), series_bins=300.0,
                                     job_ms=job_s * 1e3, sources_per_s=nsrc / job_s,
                                     stage_ms={k: v * 1e3 for k, v in stages.items()},
                                     fit_ms=m_ms, sources_per_s_fit=nsrc / (m_ms * 1e-3),
                                     note='whole job through core.run_sources, exposure tables built inside the '
                                          'timed region; stage_ms from a second run that waits after every stage')
        del srcs, mres, mcr
        torch.cuda.empty_cache()

        # Bayesian Blocks at the C1 shape: ten years of daily cells from the same photons
        bcells = data0.get_cells(tcell=86400., use_barycenter=False, quiet=True)
        bcll = core.CellsLogLikelihood(bcells)
        bcll.do_bb(prior=8)
        torch.cuda.synchronize()
        tb = []
        for _ in range(3):
            bcll.fitness_vals = None
            t0b = time.perf_counter()
            bcll._fitness_rows()
            torch.cuda.synchronize()
            t1b = time.perf_counter()
            bb = bcll.do_bb(prior=8)
            torch.cuda.synchronize()
            t2b = time.perf_counter()
            bcll.get_bb_lightcurve(bb_prior=8)
            torch.cuda.synchronize()
            tb.append((t1b - t0b, t2b - t1b, time.perf_counter() - t2b))
        tb = np.min(np.array(tb), axis=0)
        nb = len(bcells)
        extra['bayesian_blocks'] = dict(ncell=nb, fitness_rows_ms=tb[0] * 1e3, partition_ms=tb[1] * 1e3,
                                        block_lightcurve_ms=tb[2] * 1e3, blocks=int(len(bb[0])),
                                        interpolations_per_s=1000.0 * nb * (nb + 1) / 2 / tb[0])
        del bcll
        torch.cuda.empty_cache()
    if cpu_other is not None:
        extra['cpu_baselines'] = cpu_other

    line = {
        'metric': 'cells fit/s + photon×freq evals/s (1/2/4/8 GPU), % HBM/FP64 roofline',
        'value': cells_done / elapsed,
        'unit': 'cells/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / nfit * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': 'C2: %.0e photons, %g yr, %g-s cells (%d cells per list after trimming), batched '
                               'alpha-hat + TS fit; %d lists per GPU fitted in rotation, launches dealt over %d HIP '
                               'streams; every step repeated %d times inside the timed region (%d fits, %.0f ms)'
                               % (args.nph, args.years, args.tcell, ncells[0], NLIST, NSTREAM, reps, nfit,
                                  elapsed * 1e3),
                   'fits_timed': nfit, 'repeats_per_step': reps,
                   'parallelism': 'every GPU fits its own photon lists, no collective in the timed steps '
                                  '(extra.strong_c2_fit, extra.strong_c4_z2m, extra.strong_c3_fft, extra.strong_c5_sources: one job split over the GPUs, '
                                  'collectives timed)'},
        'roofline': roofline,
        'cpu_baseline': cpu,
        'extra': extra,
    }
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    run_rank(args)


if __name__ == '__main__':
    main()
