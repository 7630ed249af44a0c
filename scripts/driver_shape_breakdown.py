# coding=utf-8
"""Where the fixed cost of the driver-shaped bench call (`bench.py --steps 20 --warmup 5`) goes: the timed region of bench.py
taken apart — the epoch's sampling + batch tensor, the K steps, the final flush — each with its host time (call returns) and
its time to completion (synchronised), next to the unsplit region as bench.py times it.  Diagnostic only.

    python scripts/driver_shape_breakdown.py [--steps 20 --warmup 5 --reps 7]
"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    reps = 7
    if '--reps' in sys.argv:
        i = sys.argv.index('--reps')
        reps = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    args = bench.parse()
    if '--steps' not in ' '.join(sys.argv):
        args.steps, args.warmup = 20, 5
    from dccf_amd.models import DCCF, FusedOptimizer
    from dccf_amd.data_processor import DeviceTrainSet
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    U, I, D, F, B = args.users, args.items, args.dim, args.feat, args.batch_size
    S, A, std, p_drop, lr, l2 = 10, 2, 0.1, 0.2, 1e-3, 1e-4
    g = torch.Generator(device=dev).manual_seed(args.seed)
    feat = torch.randn(I, F, generator=g, device=dev) * 0.05
    expo = torch.empty(U, I, device=dev)
    rows = max(1, (1 << 30) // (4 * I))
    for r0 in range(0, U, rows):
        expo[r0:r0 + rows].normal_(generator=g)
    model = DCCF(path=None, dataset=None, sentence_model=None, sample_num=S, attribute_num=A, std=std, label_min=0,
                 label_max=1, feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=1,
                 random_seed=args.seed, model_path='/tmp/bench.pt', feature_embedding=feat, expo_prob=expo, ips_factors=None)
    model.apply(model.init_paras)
    opt = FusedOptimizer(model, 'adam', lr, l2)
    model.optimizer = opt
    model.train()
    n_pairs = (args.steps + args.warmup + 2) * B
    uid, iid = bench.synthetic_interactions(int(n_pairs * 1.15) + 1000, U, I, args.seed)
    ds = DeviceTrainSet(uid[:n_pairs], iid[:n_pairs], U, I, args.seed)
    model.ctx.reserve(2 * B, D, F, S, A)
    y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
    batch = {'Y': y, 'rank': 1, 'train': True, 'dropout': p_drop}

    def run(full, k0, k1):
        for k in range(k0, k1):
            batch['X'] = full[k]
            model.train_step(batch, X_next=full[k + 1] if k + 1 < k1 else None)

    def epoch_tensor(e):
        full, _ = ds.epoch_batches(e, B)
        return full[:args.warmup + args.steps].contiguous()

    full = epoch_tensor(0)
    run(full, 0, args.warmup)
    torch.cuda.synchronize()
    sync, now = torch.cuda.synchronize, time.perf_counter
    K, W = args.steps, args.warmup
    split, whole, nosamp = [], [], []
    for r in range(reps):
        # the region as bench.py times it
        t0 = now()
        full = epoch_tensor(2 * r + 1)
        run(full, W, W + K)
        opt.flush()
        sync()
        whole.append((now() - t0) * 1e3)
        # the same, taken apart
        t0 = now()
        full = epoch_tensor(2 * r + 2)
        t1 = now()
        sync()
        t2 = now()
        run(full, W, W + K)
        t3 = now()
        sync()
        t4 = now()
        opt.flush()
        t5 = now()
        sync()
        t6 = now()
        split.append([(t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3, (t5 - t4) * 1e3, (t6 - t4) * 1e3])
        # steps + flush only, queue empty at the start
        t0 = now()
        run(full, W, W + K)
        opt.flush()
        sync()
        nosamp.append((now() - t0) * 1e3)
    sp = np.median(np.array(split), axis=0)
    out = {'steps': K, 'warmup': W, 'reps': reps,
           'whole_region_ms': round(float(np.median(whole)), 4), 'whole_region_ms_per_step': round(float(np.median(whole)) / K, 5),
           'steps_plus_flush_ms': round(float(np.median(nosamp)), 4),
           'epoch_tensor_host_ms': round(float(sp[0]), 4), 'epoch_tensor_done_ms': round(float(sp[1]), 4),
           'steps_host_ms': round(float(sp[2]), 4), 'steps_done_ms': round(float(sp[3]), 4),
           'flush_host_ms': round(float(sp[4]), 4), 'flush_done_ms': round(float(sp[5]), 4),
           'all_whole_ms': [round(x, 4) for x in whole]}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
