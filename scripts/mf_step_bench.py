# coding=utf-8
"""IPSBiasedMF TRAINING throughput through the host mirror (the exposure pipeline's training): one step =
what runner._step does for the MF family — model.train_step(batch): ONE library call (mf_train_step) under the windowed lazy
regularisation; with DCCF_LAZY_K=0 model(batch) (forward + BPR + backward into the flat gradient) and optimizer.step() (explicit
l2 + clip + Adam over every parameter, row-aware) — at Electronics size, D = 64; the flush of the rows that are still behind is
inside the timed region.  One JSON line.
    python scripts/mf_step_bench.py [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dccf_amd.models import BiasedMF, FusedOptimizer
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    U, I, D = 192403, 63001, 64
    dev = torch.device('cuda', 0)
    m = BiasedMF(label_min=0, label_max=1, feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D,
                 random_seed=2019, model_path='/tmp/mf.pt')
    m.kind = 'IPSBiasedMF'                       # (IPSBiasedMF.__init__ only adds the propensity file, src/models/IPSBiasedMF.py:27)
    g = torch.Generator(device=dev).manual_seed(0)
    m.propensity, m.M = torch.rand(I, generator=g, device=dev), 0.1
    m.apply(m.init_paras)
    opt = FusedOptimizer(m, 'adam', 1e-3, 1e-4)
    m.optimizer = opt
    m.train()
    res = {}
    for B in (128, 1024, 4096):
        u = torch.randint(0, U, (steps + 10, B), generator=g, device=dev)
        X = torch.stack([torch.cat([u, u], 1), torch.randint(0, I, (steps + 10, 2 * B), generator=g, device=dev)], 2).contiguous()
        y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
        batch = {'Y': y, 'rank': 1, 'train': True, 'dropout': 0.0}

        def run(k0, k1):
            for k in range(k0, k1):
                batch['X'] = X[k]
                m.train_step(batch)
        run(0, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(10, 10 + steps)
        opt.flush()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res['batch_size=%d' % B] = {'ms_per_step': round(dt / steps * 1e3, 4), 'pairs_per_s': round(steps * B / dt),
                                    'host_ms_per_step': round((t1 - t0) / steps * 1e3, 4)}
    print(json.dumps({'metric': 'IPSBiasedMF train pairs/s (forward + BPR + backward + dense l2/clip/Adam), Electronics-shaped, D=64',
                      'regularisation': ('windowed lazy (K = %d)' % m.lazy_K) if opt.lazy is not None else 'dense pass every step',
                      'params': int(m.flat_p.numel()), 'steps': steps, 'results': res}))


if __name__ == '__main__':
    main()
