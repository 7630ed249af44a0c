# coding=utf-8
"""ORACLE — test infrastructure, NOT product code.

A CPU (numpy, fp32) restatement of the reference's DCCF / MF hot path, used only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` as the checker for the HIP path.
Nothing under ``dccf_amd/`` may import this package.

Pinned: every function here is checked in ``tests/test_oracle_golden.py`` against golden vectors obtained by
running the unmodified reference on CPU in the build container (``tests/golden/make_golden.py``); the
reference itself has no tests for this path.

All ``file:line`` citations are relative to the reference's ``src/``.
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------------
# DCCF forward / backward
# ----------------------------------------------------------------------------------------------
def expand_indices(X, sample_item, A):
    """models/DCCF.py:74-82.  Flat order l = (n*(S+1)+s)*A + a; s=0 is the true item; fid is the true item."""
    u = X[:, 0].astype(np.int64)
    i = X[:, 1].astype(np.int64)
    cand = np.concatenate([i[:, None], sample_item.astype(np.int64)], axis=1)          # [N, S+1]
    N, S1 = cand.shape
    items = np.broadcast_to(cand[:, :, None], (N, S1, A))
    users = np.broadcast_to(u[:, None, None], (N, S1, A))
    true_items = np.broadcast_to(i[:, None, None], (N, S1, A))
    return users.reshape(-1).copy(), items.reshape(-1).copy(), true_items.reshape(-1).copy(), cand


def dropout_scale(p):
    """torch dropout multiplies by keep.div(1-p) (fp32)."""
    return F32(1.0) / F32(1.0 - p) if p > 0.0 else F32(1.0)


def n_mlp_layers(P):
    """Number of mlp layers in a parameter dict (mlp.0, mlp.1, ...; src/models/DCCF.py:59-62)."""
    k = 0
    while 'mlp.%d.weight' % k in P:
        k += 1
    return k


def dccf_forward(P, feat, expo, X, sample_item, noise, keep, p, A):
    """models/DCCF.py:66-107 with the random draws injected.

    P: dict with 'uid_embeddings.weight' [U,D], 'iid_embeddings.weight' [I,D], 'mlp.0.weight' [D,D+F], 'mlp.0.bias' [D]
    and, for --n_layers > 1, 'mlp.k.weight' [D,D] / 'mlp.k.bias' [D] (src/models/DCCF.py:61-62).
    noise: [L,F] (already multiplied by std, as captured); keep: [L,D] 0/1 (one layer), [n_layers,L,D] or None;
    p: dropout probability.  Returns dict of prediction [N] and the intermediates the backward needs.
    """
    Ue, Ve = P['uid_embeddings.weight'], P['iid_embeddings.weight']
    W, b = P['mlp.0.weight'], P['mlp.0.bias']
    NL = n_mlp_layers(P)
    uid, iid, fid, cand = expand_indices(X, sample_item, A)
    N, S1 = cand.shape
    ue = Ue[uid]
    ie = Ve[iid]
    fe = feat[fid] + noise.astype(F32)
    x = np.concatenate([ie, fe], axis=1).astype(F32)
    if keep is not None:
        keep = np.asarray(keep)
        if keep.ndim == 2:
            keep = keep[None]

    def act(z, layer):       # relu + dropout (models/DCCF.py:92-94)
        r = np.maximum(z, F32(0))
        if keep is None or p == 0.0:
            km = np.ones_like(z)
        else:
            km = keep[layer].astype(F32) * dropout_scale(p)
        return (r * km).astype(F32), km

    z = (x @ W.T + b).astype(F32)
    h, km = act(z, 0)
    layers = [dict(inp=x, z=z, km=km, h=h)]
    for k in range(1, NL):
        zk = (h @ P['mlp.%d.weight' % k].T + P['mlp.%d.bias' % k]).astype(F32)
        hk, kmk = act(zk, k)
        layers.append(dict(inp=h, z=zk, km=kmk, h=hk))
        h = hk
    m = (ue * h).sum(axis=1, dtype=F32).reshape(N, S1, A)
    e = expo[uid, iid].reshape(N, S1, A).astype(F32)
    e = e - e.max(axis=1, keepdims=True)
    ee = np.exp(e, dtype=F32)
    w = (ee / ee.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
    pred = (w * m).sum(axis=1, dtype=F32).mean(axis=1, dtype=F32).astype(F32)
    return dict(prediction=pred, uid=uid, iid=iid, fid=fid, cand=cand, ue=ue, x=x, z=z, km=km, h=h, m=m, w=w, layers=layers)


def sigmoid(x):
    return F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))


def loss_and_dpred(pred, Y, rank):
    """models/DCCF.py:116-125 (same as models/BaseModel.py:210-218).  rank=1: first half positives, second half
    negatives, loss = -sum log sigmoid(pos-neg); rank=0: MSELoss (mean)."""
    N = pred.shape[0]
    if rank == 1:
        B = N // 2
        d = (pred[:B] - pred[B:]).astype(F32)
        sg = sigmoid(d)
        loss = -np.log(sg, dtype=F32).sum(dtype=F32)
        gp = -(F32(1.0) - sg)            # dL/dpos = -sigmoid(neg-pos)
        dpred = np.concatenate([gp, -gp]).astype(F32)
    else:
        diff = (pred - Y.astype(F32)).astype(F32)
        loss = (diff * diff).mean(dtype=F32)
        dpred = (F32(2.0) * diff / F32(N)).astype(F32)
    return F32(loss), dpred


def dccf_backward(P, fw, dpred, A):
    """Analytic restatement of what autograd does for models/DCCF.py:84-100 (grads of the loss term only).
    expo and feat are plain tensors (no grad); the softmax weights are constants."""
    Ue, Ve = P['uid_embeddings.weight'], P['iid_embeddings.weight']
    W = P['mlp.0.weight']
    D = Ue.shape[1]
    N, S1, _ = fw['m'].shape
    layers = fw['layers']
    dm = (dpred[:, None, None] * fw['w'] / F32(A)).astype(F32).reshape(-1)            # [L]
    due = (dm[:, None] * layers[-1]['h']).astype(F32)
    dh = (dm[:, None] * fw['ue']).astype(F32)
    out = {}
    for k in range(len(layers) - 1, 0, -1):           # the extra D -> D layers (models/DCCF.py:61-62,91-94), last to first
        ly = layers[k]
        dzk = (dh * ly['km'] * (ly['z'] > 0)).astype(F32)
        out['mlp.%d.weight' % k] = (dzk.T @ ly['inp']).astype(F32)
        out['mlp.%d.bias' % k] = dzk.sum(axis=0, dtype=F32)
        dh = (dzk @ P['mlp.%d.weight' % k]).astype(F32)
    dz = (dh * layers[0]['km'] * (layers[0]['z'] > 0)).astype(F32)
    gW = (dz.T @ fw['x']).astype(F32)
    gb = dz.sum(axis=0, dtype=F32)
    dx = (dz @ W).astype(F32)
    gU = np.zeros_like(Ue)
    gV = np.zeros_like(Ve)
    np.add.at(gU, fw['uid'], due)
    np.add.at(gV, fw['iid'], dx[:, :D])
    out.update({'uid_embeddings.weight': gU, 'iid_embeddings.weight': gV, 'mlp.0.weight': gW, 'mlp.0.bias': gb, 'dz': dz, 'dm': dm})
    return out


# ----------------------------------------------------------------------------------------------
# MF family (RecModel / BiasedMF / IPSBiasedMF)
# ----------------------------------------------------------------------------------------------
def mf_forward(P, X, kind, propensity=None, M=0.1):
    """models/RecModel.py:38-48, models/BiasedMF.py:17-33, models/IPSBiasedMF.py:37-57."""
    u, i = X[:, 0], X[:, 1]
    pu, qi = P['uid_embeddings.weight'][u], P['iid_embeddings.weight'][i]
    pred = (pu * qi).sum(axis=1, dtype=F32)
    if kind in ('BiasedMF', 'IPSBiasedMF'):
        pred = pred + P['user_bias.weight'][u, 0] + P['item_bias.weight'][i, 0] + P['global_bias']
    inv = None
    if kind == 'IPSBiasedMF':
        prop = np.maximum(propensity[i], F32(M)).astype(F32)
        pred = (pred / prop).astype(F32)
        inv = prop
    return pred.astype(F32), dict(pu=pu, qi=qi, u=u, i=i, prop=inv)


def mf_backward(P, fw, dpred, kind):
    g = dpred if fw['prop'] is None else (dpred / fw['prop']).astype(F32)   # in-place `prediction /= propensity`
    out = {k: np.zeros_like(v) for k, v in P.items()}
    np.add.at(out['uid_embeddings.weight'], fw['u'], g[:, None] * fw['qi'])
    np.add.at(out['iid_embeddings.weight'], fw['i'], g[:, None] * fw['pu'])
    if kind in ('BiasedMF', 'IPSBiasedMF'):
        np.add.at(out['user_bias.weight'][:, 0], fw['u'], g)
        np.add.at(out['item_bias.weight'][:, 0], fw['i'], g)
        out['global_bias'] = g.sum(dtype=F32)
    return out


def mf_full_matrix(P, kind, propensity=None, M=0.1):
    """README.md:28-30 'save the full predicted user-item matrix' — the reference has no code for it; this is the
    pairwise predict evaluated on every (u, i)."""
    out = (P['uid_embeddings.weight'] @ P['iid_embeddings.weight'].T).astype(F32)
    if kind in ('BiasedMF', 'IPSBiasedMF'):
        out = out + P['user_bias.weight'] + P['item_bias.weight'][:, 0][None, :] + P['global_bias']
    if kind == 'IPSBiasedMF':
        out = out / np.maximum(propensity, F32(M))[None, :]
    return out.astype(F32)


# ----------------------------------------------------------------------------------------------
# dense regularisation + clip + optimizers
# ----------------------------------------------------------------------------------------------
def l2_value(P):
    """models/BaseModel.py:179-187."""
    s = F32(0)
    for v in P.values():
        s = F32(s + (np.asarray(v, dtype=F32) ** 2).sum(dtype=F32))
    return s


def add_l2_grad(g, p, l2w):
    """autograd of l2()*l2_weight: grad += l2w * (2 * p)   (runners/BaseRunner.py:181)."""
    return (g + F32(l2w) * (F32(2.0) * p)).astype(F32)


def clip_value(g, c=50.0):
    """runners/BaseRunner.py:185."""
    return np.clip(g, F32(-c), F32(c)).astype(F32)


class DenseOptimizer(object):
    """torch.optim.{SGD,Adagrad,Adam}(lr, weight_decay=l2) with torch defaults (runners/BaseRunner.py:83-107);
    op order of torch 2.10's single-tensor CPU paths."""

    def __init__(self, name, lr, wd):
        self.name = name.lower()
        assert self.name in ('gd', 'adagrad', 'adam')
        self.lr, self.wd = lr, wd
        self.t = 0
        self.state = {}

    def step(self, P, G):
        self.t += 1
        t = self.t
        for k in P:
            p = np.asarray(P[k], dtype=F32)
            g = (np.asarray(G[k], dtype=F32) + F32(self.wd) * p).astype(F32)
            if self.name == 'gd':
                P[k] = (p + F32(-self.lr) * g).astype(F32)
            elif self.name == 'adagrad':
                st = self.state.setdefault(k, np.zeros_like(p))
                st += g * g
                std = np.sqrt(st, dtype=F32) + F32(1e-10)
                P[k] = (p + F32(-self.lr) * g / std).astype(F32)
            else:
                b1, b2, eps = 0.9, 0.999, 1e-8
                m, v = self.state.setdefault(k, (np.zeros_like(p), np.zeros_like(p)))
                m += F32(1.0 - b1) * (g - m)
                v *= F32(b2)
                v += F32(1.0 - b2) * g * g
                bc1 = 1.0 - b1 ** t
                bc2 = 1.0 - b2 ** t
                step_size = self.lr / bc1
                denom = (np.sqrt(v, dtype=F32) / F32(bc2 ** 0.5) + F32(eps)).astype(F32)
                P[k] = (p + F32(-step_size) * m / denom).astype(F32)
        return P


def train_step(P, opt, l2w, grads):
    """loss.backward() + clip + step of runners/BaseRunner.py:181-187 given the loss-term grads."""
    G = {}
    for k in P:
        G[k] = clip_value(add_l2_grad(grads[k], np.asarray(P[k], dtype=F32), l2w))
    return opt.step(P, G), G


# ----------------------------------------------------------------------------------------------
# data processor (host logic)
# ----------------------------------------------------------------------------------------------
def shuffle_in_unison(data):
    """utils/utils.py:82-92 — the same permutation for every array of the dict."""
    state = np.random.get_state()
    for k in data:
        np.random.set_state(state)
        np.random.shuffle(data[k])
    return data


def sample_neg_from_uid_list(uids, iids, neg_n, train, item_num, train_hist, vt_hist):
    """data_processor/DataProcessor.py:446-524 — consumes the global numpy RNG exactly as the reference does.
    Returns (uid_list, neg_iid_list, pos_iid_list)."""
    from collections import defaultdict
    u_out, n_out, p_out = [], [], []
    tmp = defaultdict(set)
    for idx, uid in enumerate(uids):
        if train:
            inter = train_hist.get(uid, set()) | tmp[uid]
        else:
            inter = train_hist.get(uid, set()) | vt_hist.get(uid, set()) | tmp[uid]
        remain_n = item_num - len(inter)
        remain = None
        if 1.0 * remain_n / item_num < 0.2:
            remain = [i for i in range(1, item_num) if i not in inter]
        assert remain_n >= neg_n
        if remain is None:
            for _ in range(neg_n):
                iid = np.random.randint(item_num)
                while iid in inter or iid in tmp[uid]:
                    iid = np.random.randint(item_num)
                u_out.append(uid)
                n_out.append(iid)
                tmp[uid].add(iid)
        else:
            picks = np.random.choice(remain, neg_n, replace=False)
            u_out.extend([uid] * neg_n)
            n_out.extend(picks)
            tmp[uid].update(picks)
        p_out.extend([iids[idx]] * neg_n)
        if not train:
            tmp = defaultdict(set)
    return np.array(u_out, dtype=np.int64), np.array(n_out, dtype=np.int64), np.array(p_out, dtype=np.int64)


def first_occurrence(uids, iids):
    """data_processor/DataProcessor.py:420-426 — one (uid, first iid) per distinct eval user, in order."""
    seen, fu, fi = set(), [], []
    for u, i in zip(uids, iids):
        if u not in seen:
            seen.add(u)
            fu.append(u)
            fi.append(i)
    return fu, fi


def eval_data(df, test_neg_n, item_num, train_hist, vt_hist):
    """data_processor/DataProcessor.py:73-111 + 408-444 for rank==1: positives followed by test_neg_n negatives per
    distinct user.  df: int array [n,4] (uid,iid,label,time).  Returns the data dict (uid, iid, Y, X, sample_id)."""
    fu, fi = first_occurrence(df[:, 0].tolist(), df[:, 1].tolist())
    nu, nn, _ = sample_neg_from_uid_list(fu, fi, test_neg_n, False, item_num, train_hist, vt_hist)
    uid = np.concatenate([df[:, 0], nu]).astype(np.int64)
    iid = np.concatenate([df[:, 1], nn]).astype(np.int64)
    Y = np.concatenate([df[:, 2].astype(F32), np.zeros(len(nu), dtype=F32)])
    return dict(uid=uid, iid=iid, Y=Y, X=np.stack([uid, iid], axis=1), sample_id=np.arange(len(Y)))


def train_batches(train, batch_size, item_num, train_hist):
    """data_processor/DataProcessor.py:227-250 + 160-207: one negative per train row (sampled over the whole epoch
    before the first batch), batches X=[pos;neg], Y=[1..;0..], sample_id=[ids; ids+len(train)]."""
    n = len(train['Y'])
    _, neg, _ = sample_neg_from_uid_list(train['uid'].tolist(), train['iid'].tolist(), 1, True, item_num, train_hist, {})
    out = []
    for b0 in range(0, n, batch_size):
        b1 = min(n, b0 + batch_size)
        posX = train['X'][b0:b1]
        negX = np.stack([train['uid'][b0:b1], neg[b0:b1]], axis=1)
        sid = train['sample_id'][b0:b1]
        out.append(dict(X=np.concatenate([posX, negX], 0).astype(np.int64),
                        Y=np.concatenate([np.ones(b1 - b0, F32), np.zeros(b1 - b0, F32)]),
                        sample_id=np.concatenate([sid, sid + n]), real_batch_size=b1 - b0))
    return out


def history_dicts(train_df, val_df, test_df):
    """data_loaders/DataLoader.py:167-194 + data_processor/DataProcessor.py:44-54 (label > 0 only)."""
    from collections import defaultdict
    th, vh = defaultdict(set), defaultdict(set)
    for u, i, l in train_df[:, :3]:
        if l > 0:
            th[int(u)].add(int(i))
    for df in (val_df, test_df):
        for u, i, l in df[:, :3]:
            if l > 0:
                vh[int(u)].add(int(i))
    return th, vh


# ----------------------------------------------------------------------------------------------
# metrics
# ----------------------------------------------------------------------------------------------
def dcg_at_k(r, k, method=1):
    """utils/rank_metrics.py:130-167."""
    r = np.asarray(r, dtype=np.float64)[:k]
    if r.size:
        if method == 0:
            return r[0] + np.sum(r[1:] / np.log2(np.arange(2, r.size + 1)))
        return np.sum(r / np.log2(np.arange(2, r.size + 2)))
    return 0.


def ndcg_at_k(r, k, method=1):
    """utils/rank_metrics.py:170-201."""
    dmax = dcg_at_k(sorted(r, reverse=True), k, method)
    if not dmax:
        return 0.
    return dcg_at_k(r, k, method) / dmax


def evaluate_method(p, uid, l, metrics):
    """models/BaseModel.py:55-128 (ranking metrics + rmse/mae): sort by score descending, group by uid, average."""
    p = np.asarray(p, dtype=np.float64)
    l = np.asarray(l, dtype=np.float64)
    order = np.argsort(-p, kind='stable')
    su, sl = np.asarray(uid)[order], l[order]
    o2 = np.argsort(su, kind='stable')
    su, sl = su[o2], sl[o2]
    starts = np.flatnonzero(np.r_[True, su[1:] != su[:-1]])
    groups = np.split(sl, starts[1:])
    out = []
    for metric in metrics:
        if metric == 'rmse':
            out.append(float(np.sqrt(np.mean((l - p) ** 2))))
            continue
        if metric == 'mae':
            out.append(float(np.mean(np.abs(l - p))))
            continue
        k = int(metric.split('@')[-1])
        if metric.startswith('ndcg@'):
            vals = [ndcg_at_k(g.tolist(), k, 1) for g in groups]
        elif metric.startswith('hit@'):
            vals = [int(np.sum(g[:k]) > 0) for g in groups]
        elif metric.startswith('precision@'):
            vals = []
            for g in groups:
                if len(g) < k:
                    raise ValueError('Relevance score length < k')
                vals.append(np.mean(g[:k] != 0))
        elif metric.startswith('recall@'):
            vals = [np.sum(g[:k]) / np.sum(g) for g in groups]
        elif metric.startswith('f1@'):
            vals = [2.0 * np.sum(g[:k]) / (k + np.sum(g)) for g in groups]
        else:
            raise ValueError(metric)
        out.append(float(np.average(vals)))
    return out
