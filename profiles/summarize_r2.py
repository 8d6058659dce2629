#!/usr/bin/env python3
"""Condense what profiles/collect_r2.sh gathered into one text file and two JSON files.

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
               'command': 'profiles/collect_r2.sh: rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 12 '
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
               'command': 'profiles/collect_r2.sh: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 ... -- '
                          'python3 profiles/z2m_once.py (the launches of bench.py\'s C4 leg)'}
        print('# Z^2_2 search, hardware-counted per search of 2^21 frequencies (wave instructions x 64 lanes):')
        print(json.dumps(rec))
        print()
        json.dump(rec, open(os.path.join(root, 'z2m_flops.json'), 'w'), indent=1)
    kernel_stats(os.path.join(root, 'fft_trace'), 'python3 profiles/fft_once.py (3 x power_spectrum_fft, l = 2^25)', 0.5)
    ff = counters(root, 'fft_fetch', 'rocprofv3 --pmc FETCH_SIZE (KB): profiles/fft_once.py', '')
    fw = counters(root, 'fft_write', 'rocprofv3 --pmc WRITE_SIZE (KB): profiles/fft_once.py', '')
    if ff and fw:
        fetch = sum(v['FETCH_SIZE'][0] * v['FETCH_SIZE'][1] for k, v in ff.items() if 'stockham' in k or k.startswith('ps_')) / 3
        write = sum(v['WRITE_SIZE'][0] * v['WRITE_SIZE'][1] for k, v in fw.items() if 'stockham' in k or k.startswith('ps_')) / 3
        alg = 5 * (8 * 3786912 + 16 * ((1 << 25) // 4 + 1))
        print('# power_spectrum_fft, per call: FETCH_SIZE %.0f KB (x2 on gfx950), WRITE_SIZE %.0f KB -> traffic %.3f GB; '
              'algorithmic %.3f GB; ratio %.2f' % (fetch, write, (2 * fetch + write) * 1024 / 1e9, alg / 1e9,
                                                    (2 * fetch + write) * 1024 / alg))
        print()
    kernel_stats(os.path.join(root, 'grid_trace'), 'python3 profiles/grid_once.py (3 x cell_logpdf of the C2 cells)', 0.5)
    counters(root, 'grid_sq', 'rocprofv3 --pmc SQ_*: profiles/grid_once.py', 'logpdf')
    counters(root, 'grid_write', 'rocprofv3 --pmc WRITE_SIZE (KB): profiles/grid_once.py', 'logpdf')
    bj = os.path.join(root, 'bench.json')
    if os.path.exists(bj):
        for line in open(bj):
            if line.startswith('{'):
                print('# bench line (no profiler attached): python3 bench.py')
                print(json.dumps(json.loads(line), indent=1))


if __name__ == '__main__':
    main()
