#!/usr/bin/env python3
"""Condense what profiles/collect_r3.sh gathered into one text file and two JSON files.

Usage: python3 profiles/summarize_r2.py gpurun_out/prof_TAG > profiles/NAME.txt
Also writes cell_fit_traffic.json and z2m_flops.json into the same directory."""
import collections
import csv
import glob
import json
import os
import sys


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, '**', pattern), recursive=True))


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '')


def kernel_stats(root, title, minpct=0.05):
    for f in find(root, '*kernel_stats.csv'):
        rows = list(csv.DictReader(open(f)))
        print('# rocprofv3 --kernel-trace --stats: %s' % title)
        print('%-84s %7s %12s %12s %8s' % ('kernel', 'calls', 'avg_us', 'total_ms', 'pct'))
        for r in rows:
            if float(r['Percentage']) < minpct:
                continue
            print('%-84s %7s %12.2f %12.3f %8.3f' % (short(r['Name'])[:84], r['Calls'], float(r['AverageNs']) / 1e3,
                                                    float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
        print()


def counters(root, sub, title, match):
    """{kernel: {counter: (mean per launch, launches)}} of the kernels whose name holds `match`."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in find(os.path.join(root, sub), '*counter_collection.csv'):
        for row in csv.DictReader(open(f)):
            k = short(row['Kernel_Name'])
            if match not in k:
                continue
            acc[k.split('(')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
    out = {}
    if acc:
        print('# %s (mean per launch)' % title)
    for k, d in acc.items():
        out[k] = {}
        for c, v in sorted(d.items()):
            print('%-64s %-28s launches=%-4d mean=%.6g' % (k[:64], c, len(v), sum(v) / len(v)))
            out[k][c] = (sum(v) / len(v), len(v))
    if acc:
        print()
    return out


def main():
    root = sys.argv[1]
    kernel_stats(os.path.join(root, 'trace'), 'GODOT_BENCH_STREAMS=1 python3 bench.py --steps 100 --warmup 10 --no-cpu --nfreq 16777216  '
                 '(one stream: every launch has the GPU to itself, as the HIP events of roofline.avg_launch_ms time it)')
    f = counters(root, 'pmc_fetch', 'rocprofv3 --pmc FETCH_SIZE (KB): bench.py --steps 12 --warmup 4 --no-cpu --no-extra', 'cell_fit')
    w = counters(root, 'pmc_write', 'rocprofv3 --pmc WRITE_SIZE (KB), same command', 'cell_fit')
    counters(root, 'pmc_sq', 'rocprofv3 --pmc SQ_* instruction mix of the fit kernel, same command', 'cell_fit')
    fk = [k for k in f if 'group' in k or 'lane' in k]
    if fk and fk[0] in w:
        k = fk[0]
        fetch, write = f[k]['FETCH_SIZE'][0], w[k]['WRITE_SIZE'][0]
        # gfx950: FETCH_SIZE tallies 128-B read requests at 64 B -> doubled for streaming reads
        traffic = (2 * fetch + write) * 1024
        rec = {'kernel': k + ' (C2: 1e7 photons, four lists in rotation)', 'FETCH_SIZE_KB': fetch,
               'WRITE_SIZE_KB': write, 'traffic_bytes_per_launch': int(traffic),
               'correction': 'gfx950 FETCH_SIZE tallies 128-B read requests at 64 B: doubled '
                             '(MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported',
               'command': 'profiles/collect_r3.sh: rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 12 '
                          '--warmup 4 --no-cpu --no-extra ; same with --pmc WRITE_SIZE (separate passes)'}
        print('# HBM traffic of the fit kernel per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB = %.0f bytes' % traffic)
        print(json.dumps(rec))
        print()
        json.dump(rec, open(os.path.join(root, 'cell_fit_traffic.json'), 'w'), indent=1)
    z = counters(root, 'pmc_z2m', 'rocprofv3 --pmc SQ_INSTS_VALU_*_F64: python3 profiles/z2m_once.py', 'z2m')
    zj = os.path.join(root, 'z2m_once.json')
    if z and os.path.exists(zj):
        evals = json.load(open(zj))['evals_per_launch']
        tot = collections.defaultdict(float)
        # a search may take more than one kernel per launch (partial sums + combine): add them up
        launches = max(v[1] for d in z.values() for v in d.values())
        for d in z.values():
            for c, (mean, nl) in d.items():
                tot[c] += mean * nl / 2.0          # two searches were launched
        fma, add, mul = tot.get('SQ_INSTS_VALU_FMA_F64', 0), tot.get('SQ_INSTS_VALU_ADD_F64', 0), tot.get('SQ_INSTS_VALU_MUL_F64', 0)
        rec = {'kernels': sorted(z.keys()), 'photon_freq_evals_per_search': evals,
               'fp64_fma_wave_instructions': fma, 'fp64_add_wave_instructions': add,
               'fp64_mul_wave_instructions': mul,
               'fp64_flop_per_evaluation': (2 * fma + add + mul) * 64 / evals,
               'valu_instructions_per_evaluation': tot.get('SQ_INSTS_VALU', 0) * 64 / evals,
               'command': 'profiles/collect_r3.sh: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 ... -- '
                          'python3 profiles/z2m_once.py (the launches of bench.py\'s C4 leg)'}
        print('# Z^2_2 search, hardware-counted per search of 2^21 frequencies (wave instructions x 64 lanes):')
        print(json.dumps(rec))
        print()
        json.dump(rec, open(os.path.join(root, 'z2m_flops.json'), 'w'), indent=1)
    kernel_stats(os.path.join(root, 'fft_trace'), 'python3 profiles/fft_once.py (3 x power_spectrum_fft, l = 2^25)', 0.5)
    ff = counters(root, 'fft_fetch', 'rocprofv3 --pmc FETCH_SIZE (KB): profiles/fft_once.py', '')
    fw = counters(root, 'fft_write', 'rocprofv3 --pmc WRITE_SIZE (KB): profiles/fft_once.py', '')
    if ff and fw:
        ours = lambda k: any(w in k for w in ('stockham', 'ps_', 'gather_kernel', 'columns_kernel', 'rows'))
        fetch = sum(v['FETCH_SIZE'][0] * v['FETCH_SIZE'][1] for k, v in ff.items() if ours(k)) / 3
        write = sum(v['WRITE_SIZE'][0] * v['WRITE_SIZE'][1] for k, v in fw.items() if ours(k)) / 3
        alg = 5 * (8 * 3786912 + 16 * ((1 << 25) // 4 + 1))
        print('# power_spectrum_fft, per call: FETCH_SIZE %.0f KB (x2 on gfx950), WRITE_SIZE %.0f KB -> traffic %.3f GB; '
              'algorithmic %.3f GB; ratio %.2f' % (fetch, write, (2 * fetch + write) * 1024 / 1e9, alg / 1e9,
                                                    (2 * fetch + write) * 1024 / alg))
        print()
    kernel_stats(os.path.join(root, 'grid_trace'), 'python3 profiles/grid_once.py (3 x cell_logpdf of the C2 cells)', 0.5)
    counters(root, 'grid_sq', 'rocprofv3 --pmc SQ_*: profiles/grid_once.py', 'logpdf')
    counters(root, 'grid_write', 'rocprofv3 --pmc WRITE_SIZE (KB): profiles/grid_once.py', 'logpdf')
    overlapped(root)
    kernel_stats(os.path.join(root, 'c5_trace'), 'python3 profiles/c5_full.py 256 (the multi-source job, 256 sources, '
                 'one warm-up of 8 and two timed runs)', 0.5)
    for name, title in (('c5_full.txt', 'python3 profiles/c5_full.py 1000'), ('lightcurve_time.txt', 'python3 profiles/lightcurve_time.py')):
        f = os.path.join(root, name)
        if os.path.exists(f):
            print('# ' + title)
            print(''.join(l for l in open(f) if 'amdgpu.ids' not in l))
    lines = {}
    for name in ('bench.json', 'bench_again.json'):
        bj = os.path.join(root, name)
        if os.path.exists(bj):
            for line in open(bj):
                if line.startswith('{'):
                    lines[name] = json.loads(line)
    if len(lines) == 2:
        a, b = lines['bench.json'], lines['bench_again.json']
        print('# two runs of the plain bench on this box: value %.4g and %.4g cells/s (%.2f %% apart), ms_per_step %.5f and %.5f'
              % (a['value'], b['value'], 100 * abs(a['value'] - b['value']) / a['value'], a['ms_per_step'], b['ms_per_step']))
        print()
    if 'bench.json' in lines:
        print('# bench line (no profiler attached): python3 bench.py')
        print(json.dumps(lines['bench.json'], indent=1))


def overlapped(root):
    """The fits of the timed region as the bench deals them over its streams: from the kernel trace
    of `bench.py --steps 100 --warmup 10 --no-cpu --no-extra`, the longest run of fit launches is taken
    (the timed steps; 1 000 more were launched on one stream before it) and restated per fit."""
    for f in find(os.path.join(root, 'trace3'), '*kernel_trace.csv'):
        rows = [r for r in csv.DictReader(open(f)) if 'cell_fit' in r['Kernel_Name']]
        if not rows:
            continue
        ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Stream_Id', r.get('Queue_Id', '0'))) for r in rows)
        # split into runs at gaps of more than 2 ms between a launch's start and the end of the one before
        runs, cur = [], [ev[0]]
        for e in ev[1:]:
            if e[0] - max(x[1] for x in cur[-4:]) > 2_000_000:
                runs.append(cur)
                cur = []
            cur.append(e)
        runs.append(cur)
        print('# rocprofv3 --kernel-trace: bench.py --steps 100 --warmup 10 --no-cpu --no-extra (streams as the bench deals them)')
        print('%-10s %8s %12s %14s %14s %12s %s' % ('run', 'fits', 'span_ms', 'us_per_fit', 'mean_kernel_us', 'concurrency', 'queues'))
        for i, run in enumerate(runs):
            if len(run) < 8:
                continue
            span = max(x[1] for x in run) - run[0][0]
            busy = sum(x[1] - x[0] for x in run)
            print('%-10d %8d %12.3f %14.2f %14.2f %12.2f %s' % (i, len(run), span / 1e6, span / 1e3 / len(run), busy / 1e3 / len(run),
                                                             busy / span, len(set(x[2] for x in run))))
        print()


if __name__ == '__main__':
    main()
