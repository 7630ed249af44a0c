# coding=utf-8
"""The exposure pipeline the reference describes but does not ship (README.md:28-30):

    "train IPSBiasedMF, save the full predicted user-item matrix as <dataset>.ips_expo_prob.npy"

    python -m dccf_amd.exposure --dataset Electronics --path ../datasets/ --epoch 20 --optimizer Adam --lr 0.001

1. ``<ds>.propensity.npy`` (src/models/IPSBiasedMF.py:27 loads it; nothing in the reference writes it): if missing it is
   generated from the training interactions, propensity_i = (count_i / max count)^0.5 — the popularity propensity of the
   IPS literature the model cites; an existing file is left alone.
2. IPSBiasedMF is trained through the CLI mirror (``dccf_amd.main``; every flag of the reference is passed through).
3. The full U x I prediction (``mf_predict_full``: fp32-MFMA tiles, src/models/IPSBiasedMF.py:37-57 applied to every pair)
   is written to ``<ds>.ips_expo_prob.npy`` in row blocks through a memory map, so the host never holds a second copy of
   a matrix that is 48.5 GB at Electronics size.
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch

from dccf_amd import utils


def write_propensity(path, dataset, sep=',', power=0.5):
    """propensity_i = (count_i / max count)^power over the training interactions; returns the file name."""
    import pandas as pd
    d = os.path.join(path, dataset)
    out = os.path.join(d, dataset + utils.PROPENSITY_SUFFIX)
    if os.path.exists(out):
        return out
    names = ['uid', 'iid', 'label', 'time']                            # header-less, as DataLoader reads them (DataLoader.py:79-95)
    train = pd.read_csv(os.path.join(d, dataset + utils.TRAIN_SUFFIX), sep=sep, names=names)
    frames = [train]
    for suffix in (utils.VALIDATION_SUFFIX, utils.TEST_SUFFIX):
        f = os.path.join(d, dataset + suffix)
        if os.path.exists(f):
            frames.append(pd.read_csv(f, sep=sep, names=names))
    item_num = int(max(df['iid'].max() for df in frames)) + 1            # as DataLoader counts items (DataLoader.py:134-141)
    cnt = np.bincount(train['iid'].values.astype(np.int64), minlength=item_num).astype(np.float64)
    prop = np.power(cnt / max(cnt.max(), 1.0), power).astype(np.float32)
    np.save(out, prop)
    return out


def write_full_matrix(model, out_file, rows_per_block=4096):
    """model.full_matrix() -> .npy through a memory map, one row block at a time."""
    full = model.full_matrix()                                            # [user_num, item_num] fp32 in HBM
    U, I = full.shape
    mm = np.lib.format.open_memmap(out_file, mode='w+', dtype=np.float32, shape=(U, I))
    for r0 in range(0, U, rows_per_block):
        mm[r0:r0 + rows_per_block] = full[r0:r0 + rows_per_block].cpu().numpy()
    mm.flush()
    del mm
    return out_file


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--dataset', type=str, default='ml100k-1-5')
    ap.add_argument('--path', type=str, default='../datasets/')
    ap.add_argument('--sep', type=str, default=',')
    ap.add_argument('--propensity_power', type=float, default=0.5)
    a, _ = ap.parse_known_args(argv)
    if '-h' in argv or '--help' in argv:
        print('usage: python -m dccf_amd.exposure [--dataset NAME] [--path DIR] [--sep ,] [--propensity_power 0.5] '
              '[flags of dccf_amd.main: --epoch, --optimizer, --lr, --l2, ...]\n\n'
              'Writes <path>/<dataset>/<dataset>.propensity.npy ((count_i / max count)^power over the training interactions) if it\n'
              'is missing, trains IPSBiasedMF through the CLI mirror and writes the full U x I exposure matrix\n'
              '<dataset>.ips_expo_prob.npy that DCCF loads (README.md:28-30 of the reference).')
        return None
    prop_file = write_propensity(a.path, a.dataset, sep=a.sep, power=a.propensity_power)
    from dccf_amd import main as M
    passthrough, skip = [], False
    for tok in argv:                       # --propensity_power is this tool's own flag
        if skip:
            skip = False
            continue
        if tok == '--propensity_power':
            skip = True
            continue
        passthrough.append(tok)
    if '--model_name' not in passthrough:
        passthrough += ['--model_name', 'IPSBiasedMF']
    if '--rank' not in passthrough:
        passthrough += ['--rank', '1']
    runner = M.main(passthrough)
    model = runner.model
    if os.path.exists(model.model_path):
        model.load_model()                                                # the best epoch's checkpoint (BaseRunner.py:281-283)
    out_file = os.path.join(a.path, a.dataset, a.dataset + utils.EXPO_SUFFIX)
    write_full_matrix(model, out_file)
    logging.info('propensity: %s' % prop_file)
    logging.info('exposure matrix [%d x %d] -> %s' % (model.user_num, model.item_num, out_file))
    return out_file


if __name__ == '__main__':
    main()
