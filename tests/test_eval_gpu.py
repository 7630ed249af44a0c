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


def test_cutoff_validation():
    from dccf_amd import _lib
    z = torch.zeros(4, device='cuda')
    ip = torch.tensor([0, 4], dtype=torch.int64, device='cuda')
    rows = torch.arange(4, dtype=torch.int64, device='cuda')
    with pytest.raises(RuntimeError):
        _lib.rank_eval_topk(z, z, ip, rows, [17])
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
