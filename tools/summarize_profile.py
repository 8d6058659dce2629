#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV into a short text table.

Usage: python tools/summarize_profile.py <kernel_stats.csv> [<bench.json>] > profiles/NAME.txt
"""
import csv
import json
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    print('# rocprofv3 --kernel-trace --stats summary (%s)' % sys.argv[1])
    print('%-72s %7s %12s %12s %8s' % ('kernel', 'calls', 'avg_us', 'total_ms', 'pct'))
    for r in rows:
        if float(r['Percentage']) < 1e-3:
            continue
        name = r['Name'].replace('(anonymous namespace)::', '')
        print('%-72s %7s %12.2f %12.3f %8.3f' % (name[:72], r['Calls'], float(r['AverageNs']) / 1e3,
                                                float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
    if len(sys.argv) > 2:
        for line in open(sys.argv[2]):
            line = line.strip()
            if line.startswith('{'):
                d = json.loads(line)
                print('\n# bench line of the same run')
                print(json.dumps(d, indent=1))


if __name__ == '__main__':
    main()
