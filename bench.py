#!/usr/bin/env python3
"""Benchmark of the weighted-photon likelihood hot path on MI355X.

One "step" = one batched maximum-likelihood fit (alpha-hat + TS, the
reference's get_max + get_ts) of every cell of the workload
    C2: 1e7 synthetic photons, 10 yr, 10-minute cells (zero-exposure cells
        trimmed), cells built by the GPU binning path,
with all inputs resident in HBM.  Weak scaling: every rank owns one such
photon list (seed 11 + rank) and fits its own cells: the path has no exchange
between ranks, so the timed steps hold no collective.  The per-cell records of
all ranks are gathered once over RCCL after the timed region (its time is
reported as extra.gather_ms).  Rank 0 prints one JSON line.

Also reported (rank 0, "extra"): the 200-point profile-grid rate on the same
cells, the Z^2_2 search rate (photon x frequency evaluations per second), the
power_spectrum_fft time at the C3 size (3 786 912 bins -> l = 2^25), the
one-off binning time, and the CPU baseline: the numpy oracle (a restatement
of the reference's per-cell Python path) timed on one host core on a bounded
sample of the same cells.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
FP64_PEAK_TFLOPS = 78.6    # MI355X vector FP64 peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--nph', type=int, default=10_000_000)
    ap.add_argument('--years', type=float, default=10.0)
    ap.add_argument('--tcell', type=float, default=600.0)
    ap.add_argument('--no-extra', action='store_true', help='skip the extra measurements')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    return ap.parse_args()


def event_time_ms(fn, iters, stream):
    """Average device time of fn() over iters launches, HIP events on `stream`."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(iters)]
    for a, b in evs:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))


def _cpu_sample(s, tcell):
    """Arrays of a leading slice (40 days) of the workload, small enough to hand to worker processes."""
    t_end = s['TSTART'][0] + 40 * 86400.
    m = s['ti'] < t_end
    r = s['TSTOP'] <= t_end
    return dict(ti=s['ti'][m], we=s['we'][m], TSTART=s['TSTART'][r], TSTOP=s['TSTOP'][r],
                exposure=s['exposure'][r], LIVETIME=s['LIVETIME'][r], S=s['S'], B=s['B'], E=s['E'],
                tcell=tcell)


def _cpu_worker(p, seconds):
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


def cpu_baseline(s, tcell, seconds):
    """The oracle on the host cores: one core (the reference is single-threaded), then the
    same loop in one process per core.  Must run before this process touches the GPU: the
    workers are spawned."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    p = _cpu_sample(s, tcell)
    n, dt, ncl = _cpu_worker(p, seconds)
    out = dict(value=n / dt, unit='cells/s', cores=1, kind='port',
               sample='oracle get_max+get_ts over the first %d cells (40 days) of the same '
                      'workload, repeated for %.0f s on one core' % (ncl, dt))
    cores = max(1, min(16, os.cpu_count() or 1))
    try:
        with ProcessPoolExecutor(cores, mp_context=mp.get_context('spawn')) as ex:
            res = list(ex.map(_cpu_worker, [p] * cores, [seconds / 2] * cores))
        out['all_cores'] = dict(value=sum(r[0] for r in res) / max(r[1] for r in res), unit='cells/s',
                                cores=cores,
                                sample='the same loop in %d processes, %.0f s each' % (cores, seconds / 2))
    except Exception as e:   # a box that cannot spawn workers still reports the one-core figure
        out['all_cores'] = dict(value=None, error=repr(e))
    return out


def main():
    args = parse()
    # Libraries write to stdout as well (RCCL prints a version banner when it starts):
    # file descriptor 1 points to stderr until the one JSON line is due.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    # GODOT_BENCH_FORCE_DIST=1: take the multi-rank path with a single rank too (a rehearsal
    # of the collectives on a one-GPU box)
    if world > 1 or os.environ.get('GODOT_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local))
    from godot_amd import core, ops, synth

    # ---- workload: C2, one photon list per rank ---------------------------------
    s = synth.synth(args.nph, years=args.years, seed=11 + rank)
    # CPU comparator first, while this process has not initialised the GPU (it spawns workers)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(s, args.tcell, args.cpu_seconds)

    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    stream = torch.cuda.current_stream(dev)
    data = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'],
                     device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cells = data.get_cells(tcell=args.tcell, use_barycenter=False, quiet=True)
    torch.cuda.synchronize()
    t_bin_first = time.perf_counter() - t0
    # with the running sums of the exposure table cached: mean of five calls
    t0 = time.perf_counter()
    for _ in range(5):
        cells = data.get_cells(tcell=args.tcell, use_barycenter=False, quiet=True)
    torch.cuda.synchronize()
    t_bin_again = (time.perf_counter() - t0) / 5
    cr = cells.ranges()
    ncell = cr.ncell
    nph_in = int((cr.hi - cr.lo).sum().cpu())

    def step():
        return ops.cell_fit(cr)

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()

    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([elapsed, float(ncell)], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].cpu())
    total_cells = float(tot[1].cpu())

    # The job's epilogue, outside the timed steps (the path has no exchange between ranks:
    # each fits its own cells): per-cell (alpha, TS) records of all ranks gathered once.
    gather_ms = None
    if dist is not None:
        t0 = time.perf_counter()
        rec = torch.stack([out[0], out[1]], dim=1)
        nmax = torch.tensor([ncell], device=dev)
        dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
        pad = torch.zeros((int(nmax.item()), 2), dtype=rec.dtype, device=dev)
        pad[:ncell] = rec
        gathered = torch.empty((world,) + tuple(pad.shape), dtype=rec.dtype, device=dev)
        dist.all_gather_into_tensor(gathered, pad)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3
        del gathered, pad, rec

    # ---- the second half of the metric across the ranks: one Z^2_2 search, trial
    # frequencies sharded (2^21 per rank), photons replicated, global maximum gathered
    z2_all = None
    if dist is not None and not args.no_extra:
        from godot_amd import dist as gd
        zs = synth.synth(1_000_000, years=1.0, seed=17)
        zt = torch.from_numpy(zs['ti'] - zs['ti'][0]).to(dev)
        zw = torch.from_numpy(zs['we']).to(dev)
        nf_all = (1 << 21) * world
        zspan = float(zs['ti'][-1] - zs['ti'][0])
        zf0, zdf = 1e-4, 1.0 / (4 * zspan)
        zws = torch.empty(ops._lib.lib.godot_z2m_workspace_bytes(zt.numel(), 1 << 21, 2), dtype=torch.uint8,
                          device=dev)

        def search_fn(j0, n):
            return ops.z2m(zt, zw, zf0, zdf, n, m=2, j0=j0, workspace=zws)[0]

        gd.search_sharded(nf_all, search_fn)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        zmax, jmax, _ = gd.search_sharded(nf_all, search_fn)
        torch.cuda.synchronize()
        barrier()
        zt_s = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(zt_s, op=dist.ReduceOp.MAX)
        z_s = float(zt_s[0].cpu())
        z2_all = dict(photon_freq_evals_per_s=float(zt.numel()) * nf_all / z_s, s=z_s, nph=zt.numel(),
                      nfreq=nf_all, m=2, n_gpus=world, z2_max=zmax, argmax=jmax,
                      sharding='frequencies; photons replicated; all-gather of per-rank maxima')
        del zws
        torch.cuda.empty_cache()

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (cell_fit), HIP events on its stream -----
    k_ms = event_time_ms(step, max(20, min(args.steps, 200)), stream)
    alg_bytes = 8 * nph_in + 8 * (ncell + 1) + 16 * ncell + 16 * ncell
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes (taken separately with rocprofv3 on this same
    # workload and committed under profiles/; null when the workload differs from the one profiled)
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'r1_cell_fit_traffic.json')
    if os.path.exists(tpath) and args.nph == 10_000_000 and args.tcell == 600.0 and args.years == 10.0:
        with open(tpath) as f:
            traffic = json.load(f).get('traffic_bytes_per_launch')
    roofline = dict(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic, kernel='cell_fit_group_b0_kernel<2, 16, true>',
                    algorithmic_bytes_per_launch=alg_bytes, avg_launch_ms=k_ms)

    extra = dict(ncell=ncell, photons_in_cells=nph_in,
                 binning_first_call_s=t_bin_first, binning_cached_table_s=t_bin_again)
    if z2_all is not None:
        extra['z2m_all_gpus'] = z2_all
    if gather_ms is not None:
        extra['gather_ms'] = gather_ms
    if not args.no_extra:
        # 200-point profile grid on the same cells (FP64-ALU bound)
        ops.cell_logpdf(cr, dlogl=30, npt=200)
        p_ms = event_time_ms(lambda: ops.cell_logpdf(cr, dlogl=30, npt=200), 5, stream)
        extra['profile_grid'] = dict(cells_per_s=ncell / (p_ms * 1e-3), ms=p_ms,
                                     log_evals_per_s=200.0 * nph_in / (p_ms * 1e-3))
        torch.cuda.empty_cache()
        # dense cells (the per-source shape of C1 / C5): same photons in 2-hour cells
        dcells = data.get_cells(tcell=7200., use_barycenter=False, quiet=True)
        dcr = dcells.ranges()
        ops.cell_fit(dcr)
        d_ms = event_time_ms(lambda: ops.cell_fit(dcr), 20, stream)
        dn = int((dcr.hi - dcr.lo).sum().cpu())
        extra['dense_cells'] = dict(ncell=dcr.ncell, photons_per_cell=dn / max(dcr.ncell, 1), ms=d_ms,
                                    cells_per_s=dcr.ncell / (d_ms * 1e-3),
                                    algorithmic_GBps=(8 * dn + 40 * dcr.ncell) / (d_ms * 1e-3) / 1e9)
        # Z^2_2 search: 1e6 photons x 2^21 trial frequencies (a slice of C4's 1e8)
        zs = synth.synth(1_000_000, years=1.0, seed=17)
        zt = torch.from_numpy(zs['ti'] - zs['ti'][0]).to(dev)
        zw = torch.from_numpy(zs['we']).to(dev)
        nf = 1 << 21
        span = float(zs['ti'][-1] - zs['ti'][0])
        f0, df = 1e-4, 1.0 / (4 * span)
        ws = torch.empty(ops._lib.lib.godot_z2m_workspace_bytes(zt.numel(), nf, 2), dtype=torch.uint8,
                         device=dev)
        ops.z2m(zt, zw, f0, df, nf, m=2, workspace=ws)
        z_ms = event_time_ms(lambda: ops.z2m(zt, zw, f0, df, nf, m=2, workspace=ws), 3, stream)
        evals = float(zt.numel()) * nf / (z_ms * 1e-3)
        extra['z2m'] = dict(photon_freq_evals_per_s=evals, ms=z_ms, nph=zt.numel(), nfreq=nf, m=2,
                            nominal_flops_per_eval=45, executed_valu_instr_per_eval=9.4,
                            nominal_frac_of_fp64_peak=45 * evals / (FP64_PEAK_TFLOPS * 1e12),
                            executed_frac_of_fp64_issue=2 * 9.4 * evals / (FP64_PEAK_TFLOPS * 1e12))
        del ws
        torch.cuda.empty_cache()
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
        ops.power_spectrum_fft(W, WW, S, cnt - W, cnt - 2 * W + WW, B, l)
        f_ms = event_time_ms(lambda: ops.power_spectrum_fft(W, WW, S, cnt - W, cnt - 2 * W + WW, B, l),
                             3, stream)
        extra['power_spectrum_fft'] = dict(ms=f_ms, nbins=nb, l=l)
        torch.cuda.empty_cache()

        # Bayesian Blocks at the C1 shape: ten years of daily cells from the same photons
        bcells = data.get_cells(tcell=86400., use_barycenter=False, quiet=True)
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

    line = {
        'metric': 'cells fit/s + photon×freq evals/s (1/2/4/8 GPU), % HBM/FP64 roofline',
        'value': total_cells * args.steps / elapsed,
        'unit': 'cells/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': 'C2: %.0e photons, %g yr, %g-s cells (%d cells per GPU after '
                               'trimming), batched alpha-hat + TS fit' % (
                                   args.nph, args.years, args.tcell, ncell),
                   'parallelism': 'one photon list per GPU, no collective in the timed steps; results all-gathered once afterwards'},
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


if __name__ == '__main__':
    main()
