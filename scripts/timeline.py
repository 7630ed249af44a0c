#!/usr/bin/env python3
"""Timeline of a few consecutive steps from a rocprofv3 --kernel-trace CSV: start/end (us, relative) and queue per kernel —
shows whether the side-stream optimizer pass really runs concurrently with forward/backward.
usage: timeline.py <kernel_trace.csv> [anchor kernel name] [rows]"""
import csv
import re
import sys


def main(path, anchor='k_opt_touched', n=26):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [re.match(r'(?:void )?([\w:]+)', re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])).group(1) for r in rows]
    idx = [i for i, k in enumerate(names) if k == anchor]
    i0 = max(0, idx[len(idx) // 2] - 2) if idx else max(0, len(rows) // 2)
    t0 = int(rows[i0]['Start_Timestamp'])
    for r, k in zip(rows[i0:i0 + n], names[i0:i0 + n]):
        print('%9.2f %9.2f  q%-2s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
                                        r['Queue_Id'], k))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'k_opt_touched', int(sys.argv[3]) if len(sys.argv) > 3 else 26)
