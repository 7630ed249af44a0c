# coding=utf-8
"""Ranking metrics with the reference's definitions (src/utils/rank_metrics.py:61-87,130-201 and
BaseModel.evaluate_method, src/models/BaseModel.py:55-128), vectorised over users instead of a pandas groupby loop."""
import numpy as np


def dcg_at_k(r, k, method=0):
    """src/utils/rank_metrics.py:130-167."""
    r = np.asarray(r, dtype=np.float64)[:k]
    if r.size:
        if method == 0:
            return r[0] + np.sum(r[1:] / np.log2(np.arange(2, r.size + 1)))
        elif method == 1:
            return np.sum(r / np.log2(np.arange(2, r.size + 2)))
        raise ValueError('method must be 0 or 1.')
    return 0.


def ndcg_at_k(r, k, method=0):
    """src/utils/rank_metrics.py:170-201."""
    dcg_max = dcg_at_k(sorted(r, reverse=True), k, method)
    if not dcg_max:
        return 0.
    return dcg_at_k(r, k, method) / dcg_max


def precision_at_k(r, k):
    """src/utils/rank_metrics.py:61-87."""
    assert k >= 1
    r = np.asarray(r)[:k] != 0
    if r.size != k:
        raise ValueError('Relevance score length < k')
    return np.mean(r)


def _grouped_topk(p, uid, l, kmax):
    """Labels of each user's items in descending score order, padded to kmax: -> top [n_users, kmax], sums, sizes, and
    the per-user descending-sorted labels' top-kmax (for the ideal DCG)."""
    p = np.asarray(p, dtype=np.float64)
    l = np.asarray(l, dtype=np.float64)
    uid = np.asarray(uid)
    order = np.lexsort((-p, uid))            # by user, then score descending (stable)
    su, sl = uid[order], l[order]
    starts = np.flatnonzero(np.r_[True, su[1:] != su[:-1]])
    sizes = np.diff(np.r_[starts, len(su)])
    pos = np.arange(len(su)) - np.repeat(starts, sizes)
    g = np.repeat(np.arange(len(starts)), sizes)
    top = np.zeros((len(starts), kmax))
    m = pos < kmax
    top[g[m], pos[m]] = sl[m]
    sums = np.add.reduceat(sl, starts)
    order2 = np.lexsort((-l, uid))           # ideal ordering: labels descending within a user
    il = l[order2]
    ideal = np.zeros((len(starts), kmax))
    ideal[g[m], pos[m]] = il[m]
    return top, ideal, sums, sizes


def evaluate_method(p, data, metrics):
    """src/models/BaseModel.py:55-128.  p: predictions; data: dict with 'uid' and 'Y'; metrics: list of lower-case names."""
    l = np.asarray(data['Y'], dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    ks = [int(m.split('@')[-1]) for m in metrics if '@' in m]
    grouped = _grouped_topk(p, data['uid'], l, max(ks)) if ks else None
    out = []
    for metric in metrics:
        if metric == 'rmse':
            out.append(float(np.sqrt(np.mean((l - p) ** 2))))
        elif metric == 'mae':
            out.append(float(np.mean(np.abs(l - p))))
        elif metric in ('auc', 'f1', 'accuracy', 'precision', 'recall'):
            from sklearn import metrics as skm
            fn = {'auc': skm.roc_auc_score, 'f1': skm.f1_score, 'accuracy': skm.accuracy_score,
                  'precision': skm.precision_score, 'recall': skm.recall_score}[metric]
            out.append(float(fn(l, p)))
        else:
            k = int(metric.split('@')[-1])
            top, ideal, sums, sizes = grouped
            tk, ik = top[:, :k], ideal[:, :k]
            if metric.startswith('ndcg@'):
                disc = 1.0 / np.log2(np.arange(2, k + 2))
                dcg, idcg = (tk * disc).sum(1), (ik * disc).sum(1)
                vals = np.where(idcg > 0, dcg / np.where(idcg > 0, idcg, 1.0), 0.0)
            elif metric.startswith('hit@'):
                vals = (tk.sum(1) > 0).astype(np.float64)
            elif metric.startswith('precision@'):
                if np.any(sizes < k):
                    raise ValueError('Relevance score length < k')
                vals = (tk != 0).mean(1)
            elif metric.startswith('recall@'):
                vals = tk.sum(1) / sums
            elif metric.startswith('f1@'):
                vals = 2.0 * tk.sum(1) / (k + sums)
            else:
                raise ValueError('unknown metric ' + metric)
            out.append(float(np.average(vals)))
    return out
