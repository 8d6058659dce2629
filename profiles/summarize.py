#!/usr/bin/env python3
"""Condense what profiles/collect.sh gathered into one text file.

Usage: python3 profiles/summarize.py gpurun_out/prof_TAG > profiles/NAME.txt
"""
import collections
import csv
import glob
import json
import os
import sys


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, '**', pattern), recursive=True))


def kernel_stats(root):
    for f in find(root, '*kernel_stats.csv'):
        rows = list(csv.DictReader(open(f)))
        print('# rocprofv3 --kernel-trace --stats: python3 bench.py --steps 100 --warmup 10 --no-cpu')
        print('%-84s %7s %12s %12s %8s' % ('kernel', 'calls', 'avg_us', 'total_ms', 'pct'))
        for r in rows:
            if float(r['Percentage']) < 0.05:
                continue
            name = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            print('%-84s %7s %12.2f %12.3f %8.3f' % (name[:84], r['Calls'], float(r['AverageNs']) / 1e3,
                                                    float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
        print()


def counters(root, sub, title, match='cell_fit', add=False):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in find(os.path.join(root, sub), '*counter_collection.csv'):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            if match not in k:
                continue
            acc[k.split('(')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
    out = {}
    if acc:
        print('# %s (mean per launch)' % title)
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            print('%-60s %-28s launches=%-4d mean=%.6g' % (k[:60], c, len(v), sum(v) / len(v)))
            out[c] = (out.get(c, 0.0) if add else 0.0) + sum(v) / len(v)
    if acc:
        print()
    return out


def main():
    root = sys.argv[1]
    kernel_stats(os.path.join(root, 'trace'))
    f = counters(root, 'pmc_fetch', 'rocprofv3 --pmc FETCH_SIZE (KB)')
    w = counters(root, 'pmc_write', 'rocprofv3 --pmc WRITE_SIZE (KB)')
    counters(root, 'pmc_sq', 'rocprofv3 --pmc SQ_* instruction mix')
    z = counters(root, 'pmc_z2m', 'rocprofv3 --pmc SQ_INSTS_VALU_*_F64: python3 profiles/z2m_once.py',
                 match='z2m_kernel', add=True)
    if z:
        zj = os.path.join(root, 'z2m_once.json')
        evals = json.load(open(zj))['evals'] if os.path.exists(zj) else None
        fma, add, mul = (z.get('SQ_INSTS_VALU_FMA_F64', 0), z.get('SQ_INSTS_VALU_ADD_F64', 0),
                         z.get('SQ_INSTS_VALU_MUL_F64', 0))
        print('# Z^2_m kernel, hardware-counted per launch (wave instructions x 64 lanes):')
        print(json.dumps({'fp64_fma_lane_ops': fma * 64, 'fp64_add_lane_ops': add * 64,
                          'fp64_mul_lane_ops': mul * 64, 'fp64_flops': (2 * fma + add + mul) * 64,
                          'valu_lane_instructions': z.get('SQ_INSTS_VALU', 0) * 64,
                          'photon_freq_evals': evals,
                          'fp64_flops_per_eval': (2 * fma + add + mul) * 64 / evals if evals else None,
                          'valu_instructions_per_eval': z.get('SQ_INSTS_VALU', 0) * 64 / evals if evals else None}))
        print()
    if f and w:
        fetch, write = f['FETCH_SIZE'], w['WRITE_SIZE']
        # gfx950: FETCH_SIZE tallies 128-B read requests at 64 B -> doubled for streaming reads
        traffic = (2 * fetch + write) * 1024
        print('# HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB = %.0f bytes' % traffic)
        print(json.dumps({'FETCH_SIZE_KB': fetch, 'WRITE_SIZE_KB': write,
                          'traffic_bytes_per_launch': int(traffic)}))
        print()
    bj = os.path.join(root, 'bench.json')
    if os.path.exists(bj):
        for line in open(bj):
            if line.startswith('{'):
                print('# bench line (no profiler attached): python3 bench.py')
                print(json.dumps(json.loads(line), indent=1))


if __name__ == '__main__':
    main()
