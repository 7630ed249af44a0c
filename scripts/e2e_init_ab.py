# coding=utf-8
"""The UNTRAINED model's validation NDCG@5 (what the `Init:` log line reports) per ARM and seed, on the dataset a golden names — the
evaluation path alone (evaluation negatives, candidates, noise, metric code), many seeds, seconds per seed on the GPU.  The
reference's side of the comparison is tests/golden/e2e_c1_init.npz (tests/golden/make_golden.py e2e_init: the reference's own
DataLoader / DCCF / DataProcessor / BaseRunner.evaluate, one validation pass per seed).

    python scripts/e2e_init_ab.py --golden e2e_c1 --seeds 120 --arms default,host_all,torch_draws --out gpurun_out/e2e_init_ab.json
"""
import argparse
import json
import logging
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

ARMS = {'default': (1, 1, 0), 'host_sampling': (0, 1, 0), 'host_eval': (1, 0, 0), 'host_all': (0, 0, 0), 'torch_draws': (0, 0, 1)}


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--golden', default='e2e_c1')
    p.add_argument('--seeds', type=int, default=120)
    p.add_argument('--seed0', type=int, default=3000)
    p.add_argument('--arms', default='default,host_all,torch_draws')
    p.add_argument('--out', default='gpurun_out/e2e_init_ab.json')
    a = p.parse_args()
    from dccf_amd import synth
    from dccf_amd.data_loader import DataLoader
    from dccf_amd.data_processor import DataProcessor
    from dccf_amd.models import DCCF
    from dccf_amd.runner import BaseRunner
    g = dict(np.load(os.path.join(REPO, 'tests', 'golden', a.golden + '.npz'), allow_pickle=False))
    res = {'golden': a.golden, 'metrics': 'ndcg@5,recall@5,precision@5'}
    ref_path = os.path.join(REPO, 'tests', 'golden', a.golden + '_init.npz')
    if os.path.exists(ref_path):
        r = dict(np.load(ref_path, allow_pickle=False))
        v = r['init_valid'].astype(np.float64)
        res['reference'] = {'n': int(v.shape[0]), 'mean': v.mean(0).tolist(), 'se': (v.std(0, ddof=1) / np.sqrt(v.shape[0])).tolist()}
    logging.basicConfig(level=logging.WARNING)
    tmp = tempfile.mkdtemp(prefix='e2e_init_')
    cwd = os.getcwd()
    try:
        synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', int(g['user_num']), int(g['item_num']), int(g['n_draws']),
                            feat_dim=int(g['feat_dim']), seed=int(g['data_seed']))
        os.makedirs(os.path.join(tmp, 'src'))
        os.chdir(os.path.join(tmp, 'src'))
        dl = DataLoader(path='../dataset/', dataset='toy', label='label', sep=',')
        dl.feature_info(include_id=DCCF.include_id, include_item_features=DCCF.include_item_features,
                        include_user_features=DCCF.include_user_features)
        dl.drop_neg()
        D = int(g['D'])
        for arm in a.arms.split(','):
            fused, dev_eval, torch_draws = ARMS[arm]
            vals, t0 = [], time.time()
            for k in range(a.seeds):
                seed = a.seed0 + k
                torch.manual_seed(seed)
                np.random.seed(seed)
                model = DCCF(path=dl.path, dataset=dl.dataset, sentence_model='paraphrase-distilroberta-base-v1', sample_num=10,
                             attribute_num=2, std=0.1, label_min=dl.label_min, label_max=dl.label_max, feature_num=0,
                             user_num=dl.user_num, item_num=dl.item_num, u_vector_size=D, i_vector_size=D, n_layers=1,
                             random_seed=seed, model_path=os.path.join(tmp, 'm.pt'))
                model.apply(model.init_paras)
                dp = DataProcessor(dl, model, rank=1, test_neg_n=int(g['test_neg_n']), seed=seed, fused_eval=bool(fused))
                runner = BaseRunner(optimizer='Adam', learning_rate=float(g['lr']), epoch=0, batch_size=int(g['batch_size']),
                                    eval_batch_size=128 * 128, dropout=0.2, l2=1e-4, metrics='ndcg@5,recall@5,precision@5',
                                    check_epoch=0, early_stop=1, fused_sampling=fused, device_eval=dev_eval)
                if torch_draws:
                    os.environ['DCCF_TORCH_DRAWS'] = '1'
                try:
                    vals.append([float(x) for x in runner.evaluate(model, dp.get_validation_data(), dp)])
                finally:
                    os.environ.pop('DCCF_TORCH_DRAWS', None)
                del model, dp
            v = np.asarray(vals, dtype=np.float64)
            res[arm] = {'n': int(v.shape[0]), 'mean': v.mean(0).tolist(), 'se': (v.std(0, ddof=1) / np.sqrt(v.shape[0])).tolist(),
                        'seconds': time.time() - t0, 'ndcg5_per_seed': v[:, 0].tolist()}
            line = '%s: untrained validation ndcg@5 %.5f +- %.5f (%d seeds, %.0f s)' % (arm, v[:, 0].mean(), res[arm]['se'][0], len(vals), time.time() - t0)
            if 'reference' in res:
                d = v[:, 0].mean() - res['reference']['mean'][0]
                se = float(np.sqrt(res[arm]['se'][0] ** 2 + res['reference']['se'][0] ** 2))
                res[arm]['delta_vs_reference'] = {'delta': d, 'se': se, 'delta_over_se': d / se}
                line += '; reference %.5f +- %.5f (%d seeds): delta %+.5f = %.2f se' % (res['reference']['mean'][0], res['reference']['se'][0],
                                                                                        res['reference']['n'], d, d / se)
            print(line, flush=True)
            os.chdir(cwd)
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            with open(a.out, 'w') as f:
                json.dump(res, f, indent=1)
            os.chdir(os.path.join(tmp, 'src'))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
