#!/usr/bin/env python3
"""Long-run check of the windowed lazy regularisation at the bench shape (run on the GPU box): the same 3,000 training steps
(announced, with an unannounced step, a mis-announced one and an evaluation every few hundred) with the lazy optimizer and
with the dense pass — and with the dense pass a second time, as the control: the order of the float atomics inside the backward
differs from run to run, GD carries that noise along and Adam amplifies it (an element whose gradient is noise walks +-lr per
step), so what the lazy run may differ by is what two dense runs differ by.  Rows no batch touched: bit-identical, always."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench
from dccf_amd.models import DCCF, FusedOptimizer
from dccf_amd.data_processor import DeviceTrainSet

dev = torch.device('cuda:0')
U, I, D, F, B, steps = 192403, 63001, 64, 768, 128, 3000
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.randn(I, F, generator=g, device=dev) * 0.05
ips = dict(P=torch.randn(U, 16, generator=g, device=dev) * 0.1, Q=torch.randn(I, 16, generator=g, device=dev) * 0.1,
           bu=torch.randn(U, generator=g, device=dev) * 0.1, bi=torch.randn(I, generator=g, device=dev) * 0.1,
           prop=torch.rand(I, generator=g, device=dev), b0=0.1, M=0.1)
n_pairs = (steps + 2) * B
uid, iid = bench.synthetic_interactions(int(n_pairs * 1.15) + 1000, U, I, 1)
ds = DeviceTrainSet(uid[:n_pairs], iid[:n_pairs], U, I, 1)
full, _ = ds.epoch_batches(0, B)
y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
ok = True
for opt_name, lr, l2 in (('gd', 0.01, 0.05), ('adam', 1e-3, 1e-4)):
    res = []
    for K in (0, 8, 0):
        m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=10, attribute_num=2, std=0.1, label_min=0, label_max=1,
                 feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=1, random_seed=1,
                 model_path='/tmp/s.pt', feature_embedding=feat, ips_factors=ips)
        torch.manual_seed(3)
        m.apply(m.init_paras)
        m.optimizer = FusedOptimizer(m, opt_name, lr, l2)
        m.lazy_K = K
        m.train()
        seen = torch.zeros(U, dtype=torch.bool, device=dev)
        for k in range(steps):
            X = full[k]
            seen[X[:, 0]] = True
            nxt = full[k + 1] if k + 1 < steps else None
            if k % 97 == 13:
                nxt = None                      # unannounced
            if k % 211 == 50:
                nxt = full[(k + 7) % steps]     # announced, not kept
            m.train_step({'X': X, 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.2}, X_next=nxt)
            if k % 500 == 250:
                m.eval(); m.predict({'X': full[0][:16].contiguous(), 'dropout': 0.0}); m.train()
        sd = m.state_dict()
        torch.cuda.synchronize()
        res.append(({k_: v.clone() for k_, v in sd.items()}, seen.clone()))
        del m
    (a, seen), (b, _), (c, _) = res
    never = ~seen
    ident = bool(torch.equal(a['uid_embeddings.weight'][never], b['uid_embeddings.weight'][never]))
    dmax = {k_: float((a[k_] - b[k_]).abs().max()) for k_ in a}
    dctl = {k_: float((a[k_] - c[k_]).abs().max()) for k_ in a}
    drms = {k_: (float((a[k_] - b[k_]).pow(2).mean().sqrt()), float((a[k_] - c[k_]).pow(2).mean().sqrt())) for k_ in a}
    print(opt_name, 'untouched user rows (%d) bit-identical: %s' % (int(never.sum()), ident), flush=True)
    print('   max |lazy - dense|  :', dmax, flush=True)
    print('   max |dense - dense2|:', dctl, flush=True)
    print('   rms (lazy - dense, dense - dense2):', drms, flush=True)
    # the lazy run may differ from a dense run by what two dense runs differ by (a small factor for the luck of the maximum)
    ok = ok and ident and all(drms[k_][0] <= 3.0 * drms[k_][1] + 1e-9 for k_ in a)
print('OK' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
