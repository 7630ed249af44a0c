# coding=utf-8
"""GPU: ranking metrics on the device (rank_eval_topk) against the oracle's restatement of BaseModel.evaluate_method
(src/models/BaseModel.py:55-128) and the golden vector captured from the reference (tests/golden/metrics.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dccf_oracle as O

pytestmark = pytest.mark.gpu

METRICS = ['ndcg@1', 'ndcg@5', 'ndcg@10', 'hit@5', 'precision@5', 'recall@5', 'recall@10', 'f1@5', 'hit@1']


def device_metrics(p, uid, y, metrics):
    from dccf_amd import _lib
    from dccf_amd.data_processor import DeviceEvalSet
    es = DeviceEvalSet({'X': np.zeros((len(y), 2), np.int64), 'Y': y, 'uid': uid})
    ks = sorted({int(m.split('@')[1]) for m in metrics})
    out = _lib.rank_eval_topk(torch.as_tensor(p, dtype=torch.float32).cuda(), es.Y, es.indptr, es.rows, ks).double().cpu().numpy()
    vals = []
    for m in metrics:
        name, k = m.split('@')
        j, k = ks.index(int(k)), int(k)
        col = {'ndcg': 0, 'hit': 1, 'precision': 2, 'recall': 3}.get(name)
        v = out[:, j, col] if col is not None else 2.0 * out[:, j, 2] * k / (k + out[:, len(ks), 0])
        vals.append(float(np.mean(v)))
    return vals, out


def test_golden_metrics_vector():
    g = load_golden('metrics')
    metrics = [str(m) for m in g['metrics'] if '@' in str(m)]
    for i in range(0, len(metrics), 4):      # <= 4 distinct cut-offs per call
        chunk = metrics[i:i + 4]
        mine, _ = device_metrics(g['p'], g['uid'], g['Y'], chunk)
        want = [float(g['values'][[str(m) for m in g['metrics']].index(m)]) for m in chunk]
        np.testing.assert_allclose(mine, want, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('n_users,lo,hi', [(1, 30, 31), (257, 16, 200), (5000, 16, 1100), (64, 16, 17)])
def test_ragged_groups_match_oracle(n_users, lo, hi):
    rng = np.random.RandomState(n_users)
    sizes = rng.randint(lo, hi, size=n_users)
    uid = np.repeat(rng.permutation(n_users * 3)[:n_users], sizes)
    perm = rng.permutation(len(uid))                       # rows of one user are scattered over the split
    uid = uid[perm]
    y = (rng.rand(len(uid)) < 0.03).astype(np.float32)
    # every user has at least one positive (the reference's eval sets: one positive + test_neg_n negatives)
    first = np.unique(uid, return_index=True)[1]
    y[first] = 1.0
    p = rng.permutation(len(uid)).astype(np.float32) / len(uid)   # distinct scores: no ties
    for chunk in (METRICS[:4], METRICS[4:8], METRICS[8:]):
        mine, _ = device_metrics(p, uid, y, chunk)
        want = O.evaluate_method(p.astype(np.float64), uid, y, chunk)
        np.testing.assert_allclose(mine, want, rtol=3e-6, atol=1e-7)


def test_small_groups_ties_and_no_positives():
    # user 0: 3 rows only (< k); user 1: all scores tied -> the earlier rows win; user 2: no positive -> recall is 0/0
    uid = np.array([0, 0, 0] + [1] * 20 + [2] * 20)
    y = np.zeros(len(uid), np.float32)
    y[1] = 1
    y[3 + 2] = 1          # user 1's 3rd row
    y[3 + 7] = 1          # user 1's 8th row
    p = np.r_[[0.1, 0.9, 0.5], np.full(20, 0.25), np.linspace(0, 1, 20)].astype(np.float32)
    _, out = device_metrics(p, uid, y, ['ndcg@5', 'ndcg@10'])
    # user 0: the positive is ranked first of three
    assert out[0, 0, 0] == pytest.approx(1.0) and out[0, 0, 1] == 1 and out[0, 0, 2] == pytest.approx(0.2) and out[0, 0, 3] == 1
    assert out[0, 2, 0] == 1
    # user 1 (ties -> row order): top-5 holds row 2 at rank 3; top-10 also row 7 at rank 8
    idcg = 1 + 1 / np.log2(3)
    assert out[1, 0, 0] == pytest.approx((1 / np.log2(4)) / idcg, rel=1e-6)
    assert out[1, 1, 0] == pytest.approx((1 / np.log2(4) + 1 / np.log2(9)) / idcg, rel=1e-6)
    assert out[1, 0, 3] == pytest.approx(0.5) and out[1, 1, 3] == pytest.approx(1.0)
    # user 2: no positives
    assert out[2, 0, 0] == 0 and out[2, 0, 1] == 0 and np.isnan(out[2, 0, 3]) and out[2, 2, 0] == 0


@pytest.mark.parametrize('ties', [False, True])
def test_cutoffs_beyond_16_match_oracle(ties):
    """k > 16 (src/models/BaseModel.py:83-126 takes any k): the kernel continues the ranked list in rounds of 16.  Groups
    shorter than k, groups whose size is not a multiple of 16, and — `ties` — scores with many equal values, where a round must
    resume exactly after the last popped (score, position) pair (ties go to the earlier row; the oracle's stable sort agrees)."""
    rng = np.random.RandomState(11 + ties)
    sizes = rng.randint(5, 400, size=300)
    sizes[:4] = [16, 17, 32, 100]
    uid = np.repeat(np.arange(300), sizes)            # grouped, so "earlier row" = lower index for the oracle's stable sort
    y = (rng.rand(len(uid)) < 0.1).astype(np.float32)
    y[np.unique(uid, return_index=True)[1]] = 1.0
    if ties:
        p = rng.randint(0, 7, size=len(uid)).astype(np.float32) / 7
    else:
        p = rng.permutation(len(uid)).astype(np.float32) / len(uid)
    for chunk in (['ndcg@20', 'recall@20', 'hit@17', 'f1@100'], ['ndcg@100', 'recall@50', 'ndcg@33', 'recall@400'],
                  ['ndcg@1000', 'ndcg@5', 'recall@16']):
        mine, _ = device_metrics(p, uid, y, chunk)
        want = O.evaluate_method(p.astype(np.float64), uid, y, chunk)
        np.testing.assert_allclose(mine, want, rtol=5e-6, atol=1e-7)


def test_device_auc_rmse_mae_equal_sklearn_definitions():
    """The runner's device forms of the non-ranking metrics (src/models/BaseModel.py:68-73): auc with heavily tied scores
    against a direct pair count (ties count half — sklearn's roc_auc_score), and the one-class error."""
    from dccf_amd.runner import BaseRunner
    rng = np.random.RandomState(5)
    for n, levels in ((2000, 0), (3000, 13), (50, 2)):
        y = (rng.rand(n) < 0.3).astype(np.float32)
        y[:2] = [0, 1]
        p = rng.rand(n).astype(np.float32) if not levels else (rng.randint(0, levels, n) / levels).astype(np.float32)
        pos, neg = p[y > 0].astype(np.float64), p[y == 0].astype(np.float64)
        want = ((pos[:, None] > neg[None, :]).sum() + 0.5 * (pos[:, None] == neg[None, :]).sum()) / (len(pos) * len(neg))
        got = BaseRunner._device_auc(torch.as_tensor(p).cuda(), torch.as_tensor(y).cuda())
        assert got == pytest.approx(want, abs=1e-12)
    with pytest.raises(ValueError):
        BaseRunner._device_auc(torch.rand(8).cuda(), torch.ones(8).cuda())
    assert BaseRunner._device_metrics_ok(['auc', 'rmse', 'ndcg@50']) and not BaseRunner._device_metrics_ok(['f1'])


def test_cutoff_validation():
    from dccf_amd import _lib
    z = torch.zeros(4, device='cuda')
    ip = torch.tensor([0, 4], dtype=torch.int64, device='cuda')
    rows = torch.arange(4, dtype=torch.int64, device='cuda')
    with pytest.raises(RuntimeError):
        _lib.rank_eval_topk(z, z, ip, rows, [1025])
    with pytest.raises(RuntimeError):
        _lib.rank_eval_topk(z, z, ip, rows, [0])
    with pytest.raises(RuntimeError):
        _lib.rank_eval_topk(z, z, ip, rows, [1, 2, 3, 4, 5])
    assert _lib.rank_eval_topk(z, z, ip[:1], rows, [5]).shape == (0, 2, 4)


def test_runner_device_eval_equals_host_eval(tmp_path):
    """Same model, same Philox call counter: evaluate() on the device equals the host path (numpy restatement of the
    reference's evaluate_method) — predictions are bit-identical, only the metric arithmetic differs (fp32 vs fp64)."""
    import os
    from dccf_amd import synth
    from test_e2e_gpu import run_cli
    tmp = str(tmp_path)
    synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', 300, 200, 5000, feat_dim=32, seed=3)
    r = run_cli(tmp, ['--rank', '1', '--model_name', 'DCCF', '--optimizer', 'Adam', '--lr', '0.01', '--dataset', 'toy',
                      '--path', '../dataset/', '--metric', 'ndcg@5,recall@5,precision@5,hit@10,f1@5', '--epoch', '1',
                      '--test_neg_n', '50', '--u_vector_size', '16', '--i_vector_size', '16', '--check_epoch', '0',
                      '--eval_batch_size', '1024'])
    m, dp = r.model, r.data_processor
    res = []
    for dev in (1, 0):
        r.device_eval = dev
        m._call = 1000
        res.append(r.evaluate(m, dp.get_validation_data(), dp) + r.evaluate(m, dp.get_test_data(), dp)
                   + r.evaluate(m, dp.get_train_data(-1), dp, metrics=['rmse', 'mae']))
    assert res[0][0] > 0.05
    np.testing.assert_allclose(res[0], res[1], rtol=2e-6, atol=1e-7)


def test_device_train_set_epoch_batches_layout():
    """The DEFAULT training feed (DeviceTrainSet.epoch_batches, --fused_sampling 1) on the dataset of the reference's batch
    golden (tests/golden/batches.npz): the layout rules of DataProcessor._get_feed_dict_rk / _prepare_batches_rk
    (src/data_processor/DataProcessor.py:160-207,227-250) and the negative rule of _sample_neg_from_uid_list (:446-524)."""
    from conftest import load_golden
    from dccf_amd.data_processor import DeviceTrainSet
    g = load_golden('batches')
    df = g['df/train']
    uid, iid = df[:, 0].astype(np.int64), df[:, 1].astype(np.int64)
    U_, I_, B = int(g['user_num']), int(g['item_num']), int(g['batch_size'])
    n = len(uid)
    ds = DeviceTrainSet(uid, iid, U_, I_, seed=2019)
    hist = {u: set(iid[uid == u].tolist()) for u in np.unique(uid)}
    pos_pairs = sorted(zip(uid.tolist(), iid.tolist()))
    seen_epochs = []
    for epoch in (0, 1):
        full, tail = ds.epoch_batches(epoch, B)
        nb = n // B
        # shapes: full batches [nb, 2B, 2], tail [2r, 2] with r = n mod B (the reference's last, shorter batch)
        assert tuple(full.shape) == (nb, 2 * B, 2) and full.dtype == torch.int64 and full.is_contiguous()
        r = n % B
        assert (tail is None) == (r == 0)
        if tail is not None:
            assert tuple(tail.shape) == (2 * r, 2)
        f = full.cpu().numpy()
        halves = [(f[k, :B], f[k, B:]) for k in range(nb)]
        if tail is not None:
            t = tail.cpu().numpy()
            halves.append((t[:r], t[r:]))
        P = np.concatenate([h[0] for h in halves])
        Ng = np.concatenate([h[1] for h in halves])
        # row k and row B+k of a batch carry the same uid (negatives are generated per positive row, :243-246)
        assert np.array_equal(P[:, 0], Ng[:, 0])
        # every positive of train_df exactly once per epoch
        assert sorted(zip(P[:, 0].tolist(), P[:, 1].tolist())) == pos_pairs
        # a negative is never in the user's train history and never repeated for that user within the epoch (tmp_history
        # persists over the epoch for train=True, :479,516-517); ids are valid items
        assert Ng[:, 1].min() >= 0 and Ng[:, 1].max() < I_
        for u in np.unique(uid):
            mine = Ng[Ng[:, 0] == u, 1]
            assert len(mine) == len(hist[u])
            assert len(set(mine.tolist())) == len(mine)
            assert not (set(mine.tolist()) & hist[u])
        seen_epochs.append((P.copy(), Ng.copy()))
        # same (seed, epoch) -> same epoch again (the sharded / replicated trainers rely on it)
        full2, tail2 = ds.epoch_batches(epoch, B)
        assert torch.equal(full, full2) and (tail is None or torch.equal(tail, tail2))
    # a new epoch reshuffles and redraws
    assert not np.array_equal(seen_epochs[0][0], seen_epochs[1][0])
    assert not np.array_equal(np.sort(seen_epochs[0][1][:, 1]), np.sort(seen_epochs[1][1][:, 1])) or n < 4
    # the runner's Y / sizes for these batches (BaseRunner.fit): Y = [1]*B + [0]*B, real_batch_size = B
    # (DataProcessor.py:197-206); the negatives of the kernel equal the oracle's restatement of the stream
    from oracle import philox as PH
    order = np.argsort(uid, kind='stable')
    key = np.unique(uid * I_ + iid)
    hist_indptr = np.searchsorted(key // I_, np.arange(U_ + 1)).astype(np.int64)
    ref = PH.train_negatives(2019, 1, uid, I_, hist_indptr, (key % I_).astype(np.int64))
    assert np.array_equal(ds.sample_negatives(1).cpu().numpy(), ref)


def test_device_train_set_flags_a_user_without_admissible_negative():
    """A user whose train history leaves no item to draw: the reference asserts (DataProcessor.py:495).  The device path stores
    item 0 instead of the sampler's -1 (never a negative row index in a kernel) and check_negatives() raises."""
    from dccf_amd.data_processor import DeviceTrainSet
    I_ = 12
    uid = np.concatenate([np.zeros(I_, dtype=np.int64), np.array([1, 1, 2], dtype=np.int64)])       # user 0 interacted with every item
    iid = np.concatenate([np.arange(I_, dtype=np.int64), np.array([3, 5, 7], dtype=np.int64)])
    ds = DeviceTrainSet(uid, iid, 3, I_, seed=5)
    full, tail = ds.epoch_batches(0, 4)
    X = np.concatenate([full.cpu().numpy().reshape(-1, 2), tail.cpu().numpy()])
    assert X.min() >= 0 and X[:, 1].max() < I_
    with pytest.raises(RuntimeError, match='no admissible'):
        ds.check_negatives()
    ok = DeviceTrainSet(uid[I_:], iid[I_:], 3, I_, seed=5)
    ok.epoch_batches(0, 4)
    ok.check_negatives()


def test_device_epoch_permutation_equals_the_oracle():
    """k_epoch_batches' keyed bijection (the epoch's in-unison shuffle without a sort) against oracle/philox.py::epoch_perm, which
    the CPU suite holds to a chi-square: the device evaluates the same permutation, bit for bit — sizes around powers of two, a
    tail batch, several (seed, epoch) keys."""
    from oracle import philox as PH
    from dccf_amd import _lib as L
    dev = torch.device('cuda:0')
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    for n, B, seed, epoch in ((7, 3, 2019, 0), (130, 16, 2019, 5), (1000, 128, 7, 123456), (1025, 128, 2019, 1), (4096, 100, 99, 2)):
        uid = torch.arange(n, device=dev) * 3 + 1
        iid = torch.arange(n, device=dev)                    # the positive item of row s IS s: the batches spell the permutation out
        neg = torch.arange(n, device=dev) + 10 * n
        full, tail = L.build_epoch_batches(uid, iid, neg, None, B, bad, seed, epoch)
        got = torch.cat([full[:, :B, 1].reshape(-1)] + ([tail[:tail.shape[0] // 2, 1]] if tail is not None else [])).cpu().numpy()
        ref = PH.epoch_perm(seed, epoch, n)
        assert np.array_equal(got, ref), (n, B, seed, epoch)
        negs = torch.cat([full[:, B:, 1].reshape(-1)] + ([tail[tail.shape[0] // 2:, 1]] if tail is not None else [])).cpu().numpy()
        assert np.array_equal(negs, ref + 10 * n) and int(bad) == 0
