# coding=utf-8
"""BASELINE.json config 4: IPSBiasedMF full U x I exposure-matrix predict (mf_predict_full, README.md:28-30) at a
CDs-and-Vinyl-shaped size.  out = (P Q^T + bu + bi^T + b0) / max(prop, M)^T; 2*U*I*D flop, U*I*4 B written.
Prints one JSON line: achieved TFLOP/s and write GB/s per D."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dccf_amd import _lib as L
    U, I = 75258, 64443                      # Amazon CDs-and-Vinyl 5-core order of magnitude (external recollection)
    res = {}
    g = torch.Generator(device='cuda').manual_seed(0)
    out = torch.empty((U, I), dtype=torch.float32, device='cuda')
    dims = [int(x) for x in os.environ.get('FM_DIMS', '16,32,64,128').split(',')]
    for D in dims:
        P = torch.randn(U, D, generator=g, device='cuda') * 0.1
        Q = torch.randn(I, D, generator=g, device='cuda') * 0.1
        bu, bi = torch.randn(U, generator=g, device='cuda') * 0.1, torch.randn(I, generator=g, device='cuda') * 0.1
        prop = torch.rand(I, generator=g, device='cuda')
        b0 = torch.full((1,), 0.1, device='cuda')
        m = L.mf_struct('IPSBiasedMF', P, Q, bu, bi, b0, prop, 0.1)
        for _ in range(2):
            L.mf_predict_full(m, out=out)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record()
            L.mf_predict_full(m, out=out)
            b.record()
        torch.cuda.synchronize()
        ms = min(a.elapsed_time(b) for a, b in ev)
        # rocBLAS for reference: the bare product (no bias / propensity epilogue) into the same buffer
        QTt = Q.t().contiguous()
        for _ in range(2):
            torch.mm(P, QTt, out=out)
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev2:
            a.record()
            torch.mm(P, QTt, out=out)
            b.record()
        torch.cuda.synchronize()
        mm_ms = min(a.elapsed_time(b) for a, b in ev2)
        L.mf_predict_full(m, out=out)
        torch.cuda.synchronize()
        # spot check against torch on a corner
        ref = (P[:64] @ Q[:64].T + bu[:64, None] + bi[None, :64] + 0.1) / torch.clamp(prop[:64], min=0.1)[None, :]
        err = float((out[:64, :64] - ref).abs().max())
        res['D=%d' % D] = {'ms': round(ms, 3), 'TFLOP/s': round(2.0 * U * I * D / ms / 1e9, 1),
                           'write_GB/s': round(U * I * 4 / ms / 1e6, 1), 'max_abs_err_vs_torch_corner': err,
                           'torch_mm_ms (rocBLAS, product only)': round(mm_ms, 3)}
    print(json.dumps({'metric': 'IPSBiasedMF full UxI predict', 'form': os.environ.get('DCCF_FULL_FORM', 'default'), 'users': U, 'items': I, 'out_GB': round(U * I * 4 / 1e9, 2),
                      'results': res}))


if __name__ == '__main__':
    main()
