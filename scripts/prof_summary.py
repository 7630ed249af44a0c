#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv to a readable markdown table (kernel names shortened)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?([\w:]+(?:<[^(]{0,60}>)?)', name)
    s = m.group(1) if m else name
    return s[:90]


def main(path, title=''):
    rows = list(csv.DictReader(open(path)))
    print('# rocprofv3 --kernel-trace --stats  %s\n' % title)
    print('| kernel | calls | avg us | total ms | % |')
    print('|---|---:|---:|---:|---:|')
    for r in rows[:25]:
        print('| `%s` | %s | %.2f | %.3f | %.2f |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3,
                                                    float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))


if __name__ == '__main__':
    main(sys.argv[1], ' '.join(sys.argv[2:]))
