# coding=utf-8
"""The HBM-bound embedding gather -> score -> scatter-add of the MF family (k_mf_train: RecModel / BiasedMF / IPSBiasedMF
forward + BPR + backward, src/models/BiasedMF.py:17-33, IPSBiasedMF.py:37-57, BaseModel.py:203-219) at Electronics size.
Algorithmic bytes per batch row: P[u] and Q[i] read (2 * 4D) + their gradient rows read-modify-written (2 * 2 * 4D)
+ biases (read 2 * 4, RMW 2 * 8) + ids 16 + propensity 4.  Prints one JSON line: rows/s and achieved GB/s per batch size."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dccf_amd import _lib as L
    U, I, D = 192403, 63001, 64
    g = torch.Generator(device='cuda').manual_seed(0)
    P, Q = torch.randn(U, D, generator=g, device='cuda') * 0.1, torch.randn(I, D, generator=g, device='cuda') * 0.1
    bu, bi = torch.zeros(U, device='cuda'), torch.zeros(I, device='cuda')
    b0, prop = torch.full((1,), 0.1, device='cuda'), torch.rand(I, generator=g, device='cuda')
    gP, gQ, gbu, gbi, gb0 = [torch.zeros_like(t) for t in (P, Q, bu, bi, b0)]
    m = L.mf_struct('IPSBiasedMF', P, Q, bu, bi, b0, prop, 0.1)
    ctx = L.Context(0)
    per_row = 2 * 4 * D + 2 * 2 * 4 * D + 2 * 4 + 2 * 8 + 16 + 4
    res = {}
    for B in (128, 512, 1024, 2048, 4096, 65536, 1048576):
        u = torch.randint(0, U, (B,), generator=g, device='cuda')
        X = torch.stack([torch.cat([u, u]), torch.randint(0, I, (2 * B,), generator=g, device='cuda')], 1).contiguous()
        Y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')])
        pred, loss = torch.empty(2 * B, device='cuda'), torch.zeros(1, device='cuda')
        for _ in range(3):
            L.mf_train_fwdbwd(ctx, m, X, Y, 1, gP, gQ, gbu, gbi, gb0, pred=pred, loss=loss)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record()
            L.mf_train_fwdbwd(ctx, m, X, Y, 1, gP, gQ, gbu, gbi, gb0, pred=pred, loss=loss)
            b.record()
        torch.cuda.synchronize()
        ms = min(a.elapsed_time(b) for a, b in ev)
        res['pairs=%d' % B] = {'ms': round(ms, 4), 'rows_per_s': round(2 * B / ms * 1e3), 'achieved_GBps': round(2 * B * per_row / ms / 1e6, 1),
                               'frac_of_8TBps': round(2 * B * per_row / ms / 1e6 / 8000.0, 4)}
    print(json.dumps({'metric': 'MF embedding fwd/bwd (k_mf_train, IPSBiasedMF, D=64)', 'bytes_per_row': per_row, 'results': res}))


if __name__ == '__main__':
    main()
