# coding=utf-8
"""End-to-end NDCG@5 A/B against the reference's own runs (tests/golden/e2e*.npz, written by tests/golden/make_golden.py e2e
in the build container): the CLI mirror (dccf_amd.main) is run once per seed and per ARM on the dataset the golden names,
and the per-epoch seed means +- standard errors of every arm are written next to the reference's.

    python scripts/e2e_ab.py --golden e2e --seeds 24 --arms default,host_sampling,graph,host_eval --out gpurun_out/e2e_ab.json

Arms (what differs from the reference in each is listed in profiles/r02_e2e_ab.md):
  default        fused Philox training / evaluation negatives, device shuffle, device evaluation
  host_sampling  --fused_sampling 0: the reference's numpy sampling, bit-identical batches and evaluation negatives
  graph          --use_graph 1: the step replayed as one hipGraph
  host_eval      --device_eval 0: the reference's host-side metric code
  torch_perm     DCCF_TORCH_PERM=1: the epoch's permutation by torch.randperm instead of the keyed bijection
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

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

ARMS = {
    'default': [],
    'host_sampling': ['--fused_sampling', '0'],
    'graph': ['--use_graph', '1'],
    'host_eval': ['--device_eval', '0'],
    'host_all': ['--fused_sampling', '0', '--device_eval', '0'],
    'projected': ['--eval_noise', 'projected'],
    # diagnostic: candidates / noise / dropout masks from torch's generator through the injected kernel path (no Philox)
    'torch_draws': ['--fused_sampling', '0', '--device_eval', '0'],
    # diagnostic: the epoch's permutation from torch.randperm instead of the keyed bijection of k_epoch_batches
    'torch_perm': [],
}


def stats(a):
    a = np.asarray(a, dtype=np.float64)
    return {'mean': a.mean(0).tolist(), 'se': (a.std(0, ddof=1) / np.sqrt(a.shape[0])).tolist(), 'n': int(a.shape[0])}


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--golden', default='e2e')
    p.add_argument('--seeds', type=int, default=11, help='number of seeds per arm (2019, 2020, ...)')
    p.add_argument('--arms', default='default,host_sampling')
    p.add_argument('--out', default='gpurun_out/e2e_ab.json')
    a = p.parse_args()
    from dccf_amd import synth, main as M
    g = dict(np.load(os.path.join(REPO, 'tests', 'golden', a.golden + '.npz'), allow_pickle=False))
    ref_seeds = [int(s) for s in g['seeds']]
    ref_valid = np.stack([g['seed%d/valid' % s] for s in ref_seeds])        # [seeds, epochs, metrics]
    ref_test = np.stack([g['seed%d/test' % s] for s in ref_seeds])
    ref_init = np.stack([g['seed%d/init_valid' % s] for s in ref_seeds])
    res = {'golden': a.golden, 'config': {k: int(g[k]) if np.ndim(g[k]) == 0 and k != 'lr' else None for k in
                                          ('user_num', 'item_num', 'n_draws', 'feat_dim', 'D', 'epochs', 'test_neg_n', 'batch_size')},
           'metrics': 'ndcg@5,recall@5,precision@5',
           'reference': {'valid': stats(ref_valid), 'test': stats(ref_test), 'init_valid': stats(ref_init), 'seeds': ref_seeds}}
    tmp = tempfile.mkdtemp(prefix='e2e_ab_')
    cwd = os.getcwd()
    try:
        synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', int(g['user_num']), int(g['item_num']), int(g['n_draws']),
                            feat_dim=int(g['feat_dim']), seed=int(g['data_seed']))
        os.makedirs(os.path.join(tmp, 'src'))
        os.chdir(os.path.join(tmp, 'src'))
        for arm in a.arms.split(','):
            valid, test, init = [], [], []
            t0 = time.time()
            for k in range(a.seeds):
                seed = 2019 + k
                argv = ['--rank', '1', '--model_name', 'DCCF', '--optimizer', 'Adam', '--lr', str(float(g['lr'])),
                        '--dataset', 'toy', '--path', '../dataset/', '--metric', 'ndcg@5,recall@5,precision@5',
                        '--epoch', str(int(g['epochs'])), '--test_neg_n', str(int(g['test_neg_n'])),
                        '--u_vector_size', str(int(g['D'])), '--i_vector_size', str(int(g['D'])),
                        '--random_seed', str(seed), '--batch_size', str(int(g['batch_size'])), '--check_epoch', '0',
                        '--verbose', str(logging.WARNING)] + ARMS[arm]
                if arm == 'torch_draws':
                    os.environ['DCCF_TORCH_DRAWS'] = '1'
                if arm == 'torch_perm':
                    os.environ['DCCF_TORCH_PERM'] = '1'
                try:
                    runner = M.main(argv)
                finally:
                    os.environ.pop('DCCF_TORCH_DRAWS', None)
                    os.environ.pop('DCCF_TORCH_PERM', None)
                valid.append(runner.valid_results)
                test.append(runner.test_results)
                init.append(runner.init_results[1])
            res[arm] = {'valid': stats(valid), 'test': stats(test), 'init_valid': stats(init), 'seconds': time.time() - t0,
                        'flags': ARMS[arm], 'valid_ndcg5_per_seed': np.asarray(valid)[:, :, 0].tolist()}
            # the seeds both sides ran: with --fused_sampling 0 the batches and evaluation negatives of a seed are the reference's own,
            # so the per-seed differences are paired
            common = [k for k in range(a.seeds) if 2019 + k in ref_seeds]
            if common:
                dv = np.array([np.asarray(valid[k])[:, 0] - g['seed%d/valid' % (2019 + k)][:, 0] for k in common])
                res[arm]['paired'] = {'n': len(common), 'mean_diff': dv.mean(0).tolist(),
                                      'se_diff': (dv.std(0, ddof=1) / np.sqrt(len(common))).tolist()}
                print(arm, 'paired over', len(common), 'common seeds: mean diff', np.round(dv.mean(0), 4), 'se',
                      np.round(dv.std(0, ddof=1) / np.sqrt(len(common)), 4))
            di = np.array(res[arm]['init_valid']['mean'])[0] - np.array(res['reference']['init_valid']['mean'])[0]
            sei = np.sqrt(np.array(res[arm]['init_valid']['se'])[0] ** 2 + np.array(res['reference']['init_valid']['se'])[0] ** 2)
            print(arm, 'init valid ndcg@5: mine %.4f ref %.4f delta/se %.2f' % (res[arm]['init_valid']['mean'][0],
                                                                                  res['reference']['init_valid']['mean'][0], di / sei))
            d = np.array(res[arm]['valid']['mean'])[:, 0] - np.array(res['reference']['valid']['mean'])[:, 0]
            se = np.sqrt(np.array(res[arm]['valid']['se'])[:, 0] ** 2 + np.array(res['reference']['valid']['se'])[:, 0] ** 2)
            res[arm]['valid_ndcg5_delta'] = d.tolist()
            res[arm]['valid_ndcg5_delta_over_se'] = (d / se).tolist()
            print(arm, 'valid ndcg@5 per epoch: mine', np.round(res[arm]['valid']['mean'], 4)[:, 0], '+-',
                  np.round(res[arm]['valid']['se'], 4)[:, 0], 'ref', np.round(res['reference']['valid']['mean'], 4)[:, 0], '+-',
                  np.round(res['reference']['valid']['se'], 4)[:, 0], 'delta/se', np.round(d / se, 2), flush=True)
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
