# coding=utf-8
"""Forward / backward kernel time at the bench shape in FUSED mode (noise drawn in registers, regenerated in the backward)
against INJECTED mode (noise read from a [L, F] tensor): the upper bound of what a backward that LOADS the forward's noise
instead of regenerating it can gain.  Event-bracketed per kernel (ctx.profile), queue kept full."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--batch_sizes', default='128,512,4096')
    p.add_argument('--dim', type=int, default=64)
    p.add_argument('--feat', type=int, default=768)
    p.add_argument('--reps', type=int, default=50)
    a = p.parse_args()
    from dccf_amd import _lib as L
    dev = torch.device('cuda:0')
    U, I, D, F, S, A = 192403, 63001, a.dim, a.feat, 10, 2
    g = torch.Generator(device=dev).manual_seed(1)
    P = [torch.randn(U, D, generator=g, device=dev) * 0.01, torch.randn(I, D, generator=g, device=dev) * 0.01,
         torch.randn(D, D + F, generator=g, device=dev) * 0.01, torch.randn(D, generator=g, device=dev) * 0.01]
    feat = torch.randn(I, F, generator=g, device=dev) * 0.05
    ips = dict(P=torch.randn(U, 16, generator=g, device=dev) * 0.1, Q=torch.randn(I, 16, generator=g, device=dev) * 0.1,
               bu=torch.randn(U, generator=g, device=dev) * 0.1, bi=torch.randn(I, generator=g, device=dev) * 0.1,
               prop=torch.rand(I, generator=g, device=dev), b0=0.1, M=0.1)
    m = L.model_struct(P[0], P[1], P[2], P[3], feat, None, S, A, 0.1, ips=ips)
    G = [torch.zeros_like(x) for x in P]
    ctx = L.Context(0)
    out = {}
    for B in [int(x) for x in a.batch_sizes.split(',')]:
        N = 2 * B
        Ld = N * (S + 1) * A
        u = torch.randint(0, U, (B,), generator=g, device=dev)
        X = torch.cat([torch.stack([u, torch.randint(0, I, (B,), generator=g, device=dev)], 1),
                       torch.stack([u, torch.randint(0, I, (B,), generator=g, device=dev)], 1)]).contiguous()
        Y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
        si = L.debug_candidates(N, S, I, 7, 1, dev)
        nz = L.debug_noise(Ld, F, 0.1, 7, 1, dev)
        kp = L.debug_keep(Ld, D, 0.2, 7, 1, dev)
        modes = {'fused': L.rand_struct(seed=7, step=1), 'injected': L.rand_struct(sample_item=si, noise=nz, keep=kp)}
        for name, r in modes.items():
            for _ in range(5):
                L.dccf_train_fwdbwd(ctx, m, r, X, Y, 1, 0.2, *G)
            torch.cuda.synchronize()
            blocker = torch.zeros(128 << 20, device=dev)
            for _ in range(32):
                blocker.add_(1.0)
            ctx.profile(True)
            for _ in range(a.reps):
                L.dccf_train_fwdbwd(ctx, m, r, X, Y, 1, 0.2, *G)
            torch.cuda.synchronize()
            pr = ctx.profile_read()
            ctx.profile(False)
            out['B%d_%s' % (B, name)] = {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in pr.items()}
            print(B, name, out['B%d_%s' % (B, name)], flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
