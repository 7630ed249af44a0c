# coding=utf-8
"""Evaluation pass at Electronics scale: `--eval_users` users x (1 + test_neg_n) candidate rows, batched
predict (eval_batch_size rows, reference default 128*128) + rank_eval_topk, everything resident on the GPU.
Prints one JSON line: eval rows/s and the split of predict vs ranking time."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--users', type=int, default=192403)
    p.add_argument('--items', type=int, default=63001)
    p.add_argument('--dim', type=int, default=64)
    p.add_argument('--feat', type=int, default=768)
    p.add_argument('--eval_users', type=int, default=20000)
    p.add_argument('--test_neg_n', type=int, default=1000)
    p.add_argument('--eval_batch_size', type=int, default=128 * 128)
    p.add_argument('--reps', type=int, default=2)
    p.add_argument('--eval_noise', type=str, default='full', help='full | projected (see dccf_predict_projected)')
    a = p.parse_args()
    from dccf_amd import _lib
    from dccf_amd.models import DCCF
    from dccf_amd.runner import BaseRunner
    from dccf_amd.data_processor import DeviceEvalSet
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(1)
    feat = torch.randn(a.items, a.feat, generator=g, device=dev) * 0.05
    ips = dict(P=torch.randn(a.users, 64, generator=g, device=dev) * 0.1, Q=torch.randn(a.items, 64, generator=g, device=dev) * 0.1,
               bu=torch.randn(a.users, generator=g, device=dev) * 0.1, bi=torch.randn(a.items, generator=g, device=dev) * 0.1,
               prop=torch.rand(a.items, generator=g, device=dev), b0=0.1, M=0.1)
    m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=10, attribute_num=2, std=0.1, label_min=0, label_max=1,
             feature_num=0, user_num=a.users, item_num=a.items, u_vector_size=a.dim, i_vector_size=a.dim, n_layers=1,
             random_seed=1, model_path='/tmp/e.pt', feature_embedding=feat, ips_factors=ips)
    m.apply(m.init_paras)
    m.eval_noise = a.eval_noise
    rng = np.random.RandomState(0)
    per = 1 + a.test_neg_n
    users = rng.choice(a.users, a.eval_users, replace=False)
    uid = np.concatenate([users, np.repeat(users, a.test_neg_n)])          # positives first, then negatives (DataProcessor.py:73-111)
    iid = rng.randint(0, a.items, size=len(uid))
    y = np.r_[np.ones(a.eval_users, np.float32), np.zeros(a.eval_users * a.test_neg_n, np.float32)]
    data = {'X': np.stack([uid, iid], 1), 'Y': y, 'uid': uid}

    class DP(object):
        es = None

        def device_eval_set(self, d):
            if self.es is None:
                self.es = DeviceEvalSet(d)
            return self.es
    dp = DP()
    r = BaseRunner(optimizer='Adam', metrics='ndcg@5,recall@5,precision@5', eval_batch_size=a.eval_batch_size)
    t0 = time.time()
    r.evaluate(m, data, dp)
    torch.cuda.synchronize()
    t_first = time.time() - t0
    t0 = time.time()
    for _ in range(a.reps):
        res = r.evaluate(m, data, dp)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.reps
    es = dp.es
    pred = torch.rand(es.n, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        _lib.rank_eval_topk(pred, es.Y, es.indptr, es.rows, [5])
    torch.cuda.synchronize()
    t_rank = (time.time() - t0) / 5
    print(json.dumps({'metric': 'eval_rows_per_s', 'value': es.n / dt, 'rows': es.n, 'users': a.eval_users, 'per_user': per,
                      'eval_s': dt, 'first_eval_s_incl_upload_and_csr': t_first, 'rank_kernel_s': t_rank,
                      'full_split_estimate_s': dt * a.users / a.eval_users, 'eval_noise': a.eval_noise, 'result': res}))


if __name__ == '__main__':
    main()
