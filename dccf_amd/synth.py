# coding=utf-8
"""Synthetic, Electronics-shaped datasets in the reference's on-disk formats.

The reference ships no data and no generators for its side files, so every
configuration of BASELINE.json runs on data produced here:

  <path>/<ds>/<ds>.train.csv / .validation.csv / .test.csv   header-less ``uid,iid,label,time``
        (read by reference ``src/data_loaders/DataLoader.py:79-98``)
  <path>/<ds>/<ds>_<sentence_model>.npy        [I, F] fp32   (``src/models/DCCF.py:55``)
  <path>/<ds>/<ds>.ips_expo_prob.npy           [U, I] fp32   (``src/models/DCCF.py:64``)
  <path>/<ds>/<ds>.propensity.npy              [I]    fp32   (``src/models/IPSBiasedMF.py:27``)

The split follows ``src/data_preprocessing/amazon_data_split_RAND.py:68-93,120-122``: per user a random
20 % of the interactions go to test, 10 % to validation, the rest to train (users with few interactions
keep everything in train).
"""
import os

import numpy as np

SENTENCE_MODEL = 'paraphrase-distilroberta-base-v1'


def make_interactions(user_num, item_num, n_draws, seed=7, zipf_a=0.8):
    """user ~ Uniform, item ~ Zipf(zipf_a) over a random item permutation; de-duplicated.

    Every user and every item id in [0, num) appears at least once (so ``column_max + 1`` recovers the
    counts the way ``DataLoader._load_info`` computes them, ``src/data_loaders/DataLoader.py:141-147``).
    """
    rng = np.random.RandomState(seed)
    w = 1.0 / np.power(np.arange(1, item_num + 1, dtype=np.float64), zipf_a)
    w /= w.sum()
    perm = rng.permutation(item_num)
    uids = rng.randint(0, user_num, size=n_draws).astype(np.int64)
    iids = perm[rng.choice(item_num, size=n_draws, p=w)].astype(np.int64)
    # coverage rows: one per user and one per item
    uids = np.concatenate([uids, np.arange(user_num), rng.randint(0, user_num, size=item_num)])
    iids = np.concatenate([iids, rng.randint(0, item_num, size=user_num), np.arange(item_num)])
    key = uids * item_num + iids
    _, first = np.unique(key, return_index=True)
    first.sort()
    uids, iids = uids[first], iids[first]
    times = rng.randint(1_000_000_000, 1_500_000_000, size=len(uids)).astype(np.int64)
    return uids, iids, times


def split_rand(uids, iids, times, seed=7, test_frac=0.2, valid_frac=0.1, min_inter=5):
    """Per-user random leave-out split (train / validation / test masks)."""
    rng = np.random.RandomState(seed + 1)
    order = np.lexsort((rng.rand(len(uids)), uids))
    su = uids[order]
    starts = np.flatnonzero(np.r_[True, su[1:] != su[:-1]])
    ends = np.r_[starts[1:], len(su)]
    which = np.zeros(len(uids), dtype=np.int8)  # 0 train, 1 validation, 2 test
    for s, e in zip(starts, ends):
        n = e - s
        if n < min_inter:
            continue
        n_test = max(1, int(n * test_frac))
        n_val = max(1, int(n * valid_frac))
        which[order[s:s + n_test]] = 2
        which[order[s + n_test:s + n_test + n_val]] = 1
    return which


def write_dataset(path, dataset, user_num, item_num, n_draws, feat_dim=768, seed=7,
                  expo='random', sentence_model=SENTENCE_MODEL, feat_std=0.05, write_expo=True):
    """Writes one dataset directory. Returns a dict of the arrays written (for tests)."""
    d = os.path.join(path, dataset)
    os.makedirs(d, exist_ok=True)
    uids, iids, times = make_interactions(user_num, item_num, n_draws, seed=seed)
    which = split_rand(uids, iids, times, seed=seed)
    out = {}
    for code, suffix in ((0, '.train.csv'), (1, '.validation.csv'), (2, '.test.csv')):
        m = which == code
        arr = np.stack([uids[m], iids[m], np.ones(m.sum(), dtype=np.int64), times[m]], axis=1)
        np.savetxt(os.path.join(d, dataset + suffix), arr, fmt='%d', delimiter=',')
        out[suffix] = arr
    rng = np.random.RandomState(seed + 2)
    feat = (rng.randn(item_num, feat_dim) * feat_std).astype(np.float32)
    np.save(os.path.join(d, '%s_%s.npy' % (dataset, sentence_model)), feat)
    cnt = np.bincount(iids[which == 0], minlength=item_num).astype(np.float64)
    prop = np.power(np.maximum(cnt, 1.0) / max(cnt.max(), 1.0), 0.5).astype(np.float32)
    np.save(os.path.join(d, dataset + '.propensity.npy'), prop)
    out.update(feat=feat, propensity=prop)
    if write_expo:
        if isinstance(expo, np.ndarray):
            ex = expo.astype(np.float32)
        else:
            ex = rng.randn(user_num, item_num).astype(np.float32)
        np.save(os.path.join(d, dataset + '.ips_expo_prob.npy'), ex)
        out['expo'] = ex
    return out
