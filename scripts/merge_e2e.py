# coding=utf-8
"""Merges per-seed outputs of `tests/golden/make_golden.py e2e` (run in parallel processes, one or a few seeds each) into one
fixture:  python scripts/merge_e2e.py tests/golden/e2e.npz part1.npz part2.npz ...   (an existing target is merged too)."""
import os
import sys

import numpy as np

out = sys.argv[1]
parts = ([out] if os.path.exists(out) else []) + sys.argv[2:]
rec, seeds = {}, []
for p in parts:
    g = dict(np.load(p, allow_pickle=False))
    for k, v in g.items():
        if k.startswith('seed') and '/' in k:
            rec[k] = v
        elif k not in ('seeds', 'threads'):
            if k in rec and k != 'lr':
                assert np.array_equal(rec[k], v), (k, rec[k], v)
            rec[k] = v
    seeds += [int(s) for s in g['seeds']]
rec['seeds'] = np.array(sorted(set(seeds)))
np.savez_compressed(out, **rec)
print(out, 'seeds', rec['seeds'])
