# coding=utf-8
"""End-to-end CLI run at a non-toy size: writes a synthetic dataset in the reference's on-disk format (dccf_amd/synth.py) and
runs the reference's README command through the CLI mirror (dccf_amd.main) — data loading, evaluation negatives on the GPU,
per-epoch training (fused negatives), on-device evaluation of train/validation/test with --test_neg_n 1000, checkpointing.
Prints one JSON line with the wall-clock of each phase."""
import argparse
import json
import os
import re
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--users', type=int, default=50000)
    p.add_argument('--items', type=int, default=20000)
    p.add_argument('--draws', type=int, default=700000)
    p.add_argument('--epoch', type=int, default=2)
    p.add_argument('--batch_size', type=int, default=128)
    a = p.parse_args()
    from dccf_amd import synth, main as M
    tmp = tempfile.mkdtemp(prefix='dccf_cli_')
    try:
        t0 = time.time()
        big = a.users * a.items > (1 << 31)
        synth.write_dataset(os.path.join(tmp, 'dataset'), 'syn', a.users, a.items, a.draws, feat_dim=768, seed=11,
                            write_expo=not big)
        if big:      # a 48.5 GB exposure file: drawn on the GPU, written through a memory map in 1 GiB slabs
            import numpy as np
            import torch
            f = os.path.join(tmp, 'dataset', 'syn', 'syn.ips_expo_prob.npy')
            mm = np.lib.format.open_memmap(f, mode='w+', dtype=np.float32, shape=(a.users, a.items))
            g = torch.Generator(device='cuda').manual_seed(13)
            rows = max(1, (1 << 30) // (4 * a.items))
            for r0 in range(0, a.users, rows):
                n = min(rows, a.users - r0)
                mm[r0:r0 + n] = torch.randn(n, a.items, generator=g, device='cuda').cpu().numpy()
            mm.flush()
            del mm
        t_data = time.time() - t0
        os.makedirs(os.path.join(tmp, 'src'))
        os.chdir(os.path.join(tmp, 'src'))
        t0 = time.time()
        runner = M.main(['--rank', '1', '--model_name', 'DCCF', '--optimizer', 'Adam', '--lr', '0.001', '--dataset', 'syn',
                         '--path', '../dataset/', '--metric', 'ndcg@5,recall@5,precision@5', '--epoch', str(a.epoch),
                         '--test_neg_n', '1000', '--batch_size', str(a.batch_size), '--check_epoch', '0'])
        t_run = time.time() - t0
        logf = [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(tmp, 'log')) for f in fs][0]
        txt = open(logf).read()
        ep = re.findall(r'Epoch\s+(\d+) \[([\d.]+) s\]\s+train= ([\d.,-]+) validation= ([\d.,-]+) test= ([\d.,-]+) \[([\d.]+) s\]', txt)
        n_train = int(re.search(r'size of train: (\d+)', txt).group(1))
        n_val, n_test = [int(re.search(r'size of %s: (\d+)' % k, txt).group(1)) for k in ('validation', 'test')]
        out = {'users': a.users, 'items': a.items, 'train_rows': n_train, 'validation_rows': n_val, 'test_rows': n_test,
               'batch_size': a.batch_size, 'write_dataset_s': round(t_data, 1), 'cli_total_s': round(t_run, 1),
               'epochs': [{'epoch': int(e[0]), 'train_s': float(e[1]), 'train_pairs_per_s': round(n_train / max(float(e[1]), 1e-9)),
                           'eval_s': float(e[5]), 'valid_ndcg@5': float(e[3].split(',')[0])} for e in ep]}
        print(json.dumps(out))
    finally:
        os.chdir('/')
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
