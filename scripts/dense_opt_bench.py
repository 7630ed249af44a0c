#!/usr/bin/env python3
"""The dense regularised Adam pass (k_dense_opt_rows) on a buffer far beyond the 256 MiB Infinity Cache — 268 M parameters, p + m + v
= 3.2 GB, what bench.py reports as roofline.frac_beyond_llc — swept over the launch knobs: float4 triples in flight per lane
(DCCF_OPT_UN), non-temporal loads / stores (DCCF_OPT_NT), workgroup count (DCCF_OPT_GRID).  On the GPU box:
    DCCF_OPT_TUNE=1 python scripts/dense_opt_bench.py > gpurun_out/dense_opt_bench.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['DCCF_OPT_TUNE'] = '1'
from dccf_amd import _lib as L      # noqa: E402


def main():
    dev = torch.device('cuda:0')
    n = int(os.environ.get('BENCH_PARAMS', str(268435456)))
    D = 64
    p = torch.randn(n, device=dev) * 0.01
    g = torch.zeros(n, device=dev)
    s1, s2 = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    flags = torch.zeros(n // D, dtype=torch.uint8, device=dev)
    segs = [(0, n // D, D, flags)]
    res = []
    os.environ['DCCF_OPT_BIG_N'] = '1'          # (every size takes the knob-driven launch in this sweep)
    full = (n + 3) // 4 // 256 + 1             # one float4 slot per thread
    grids = sorted({g for g in (4096, 16384, 65536, full) if g <= full})
    for un in (1, 2, 4):
        for nt in (0, 1):
            for grid in grids:
                os.environ.update(DCCF_OPT_UN=str(un), DCCF_OPT_NT=str(nt), DCCF_OPT_GRID=str(grid))
                ev = []
                for k in range(14):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    L.dense_opt_step_rows('adam', p, g, s1, s2, 1e-3, 1e-4, 1e-4, 50.0, k + 1, segs)
                    b.record()
                    ev.append((a, b))
                torch.cuda.synchronize()
                ms = sorted(a.elapsed_time(b) for a, b in ev[2:])
                med = ms[len(ms) // 2]
                res.append({'un': un, 'nt': nt, 'grid': grid, 'ms_median': round(med, 4), 'ms_min': round(ms[0], 4),
                            'TBps_median': round(24.0 * n / 1e12 / (med / 1e3), 3), 'frac_of_8TBps': round(24.0 * n / 1e12 / (med / 1e3) / 8.0, 4)})
                print(json.dumps(res[-1]), file=sys.stderr, flush=True)
    best = max(res, key=lambda r: r['TBps_median'])
    print(json.dumps({'params': n, 'bytes_per_param': 24, 'best': best, 'all': res}))


if __name__ == '__main__':
    main()
