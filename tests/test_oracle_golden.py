# coding=utf-8
"""Pins the oracle (oracle/) against golden vectors captured from the unmodified reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import dccf_oracle as O

DCCF_CASES = ['dccf_d16_f32_adam', 'dccf_d64_f768_adam', 'dccf_d64_f32_nodrop_gd', 'dccf_d32_f160_adagrad',
              'dccf_d128_f768_adam', 'dccf_d64_f768_mse',
              # round 2 (make_golden.py dccf_ext): --n_layers 2 / 3, widths that are not a kernel tile, F > 896
              'dccf_d64_f768_l2_adam', 'dccf_d24_f100_l3_adagrad', 'dccf_d48_f1024_adam', 'dccf_d100_f800_gd',
              'dccf_d128_f32_l2_mse',
              # round 3: embedding sizes above 128 (src/models/RecModel.py:17-27 accepts any)
              'dccf_d192_f768_adam', 'dccf_d256_f96_adagrad', 'dccf_d160_f1000_gd_mse']


def pkeys(g):
    """state_dict keys of the case, in the reference's order (mlp.k.* for k < n_layers after the embeddings)."""
    return [k[5:] for k in g if k.startswith('init/')]

# fp32 tolerances: the oracle sums in numpy/BLAS order, the reference in ATen order.
FWD_RTOL, FWD_ATOL = 2e-5, 1e-6
GRAD_RTOL, GRAD_ATOL = 1e-4, 2e-6


def close(a, b, rtol, atol, what=''):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * scale, '%s: max err %g (scale %g)' % (what, err, scale)


@pytest.mark.parametrize('name', DCCF_CASES)
def test_index_expansion_bit_exact(name):
    g = load_golden(name)
    A = int(g['A'])
    X, si = g['s0/X'], g['s0/sample_item']
    uid, iid, fid, cand = O.expand_indices(X, si, A)
    N, S = si.shape
    L = N * (S + 1) * A
    assert uid.shape == (L,) and uid.dtype == np.int64
    l = np.arange(L)
    n, s = l // ((S + 1) * A), (l // A) % (S + 1)
    assert np.array_equal(uid, X[n, 0])
    assert np.array_equal(fid, X[n, 1])
    assert np.array_equal(iid, np.where(s == 0, X[n, 1], si[n, np.maximum(s - 1, 0)]))
    assert g['s0/noise'].shape == (L, int(g['F']))


@pytest.mark.parametrize('name', DCCF_CASES)
def test_dccf_train_steps_match_reference(name):
    g = load_golden(name)
    A, p, rank = int(g['A']), float(g['dropout']), int(g['rank'])
    PKEYS = pkeys(g)
    assert len(PKEYS) == 2 + 2 * int(g.get('n_layers', 1))
    P = {k: g['init/' + k].copy() for k in PKEYS}
    opt = O.DenseOptimizer(str(g['optimizer']), float(g['lr']), float(g['l2']))
    for s in range(int(g['steps'])):
        pre = 's%d/' % s
        fw = O.dccf_forward(P, g['feat'], g['expo'], g[pre + 'X'], g[pre + 'sample_item'], g[pre + 'noise'],
                            g[pre + 'mask'], p, A)
        close(fw['prediction'], g[pre + 'prediction'], FWD_RTOL, FWD_ATOL, name + ' pred')
        loss, dpred = O.loss_and_dpred(fw['prediction'], g[pre + 'Y'], rank)
        close(loss, g[pre + 'loss'], FWD_RTOL, FWD_ATOL, name + ' loss')
        close(O.l2_value(P), g[pre + 'l2'], 1e-5, 0, name + ' l2')
        grads = O.dccf_backward(P, fw, dpred, A)
        for k in PKEYS:
            close(grads[k], g[pre + 'gloss/' + k], GRAD_RTOL, GRAD_ATOL, name + ' gloss ' + k)
            gpre = O.add_l2_grad(grads[k], P[k], float(g['l2']))
            close(gpre, g[pre + 'gpre/' + k], GRAD_RTOL, GRAD_ATOL, name + ' gpre ' + k)
            close(O.clip_value(gpre), g[pre + 'gpost/' + k], GRAD_RTOL, GRAD_ATOL, name + ' gpost ' + k)
        # dense regularisation: every row of both tables has a non-zero grad
        assert np.all(np.abs(g[pre + 'gpre/uid_embeddings.weight']).sum(1) > 0)
        P, _ = O.train_step(P, opt, float(g['l2']), grads)
        for k in PKEYS:
            close(P[k], g[pre + 'after/' + k], 2e-5, 1e-7, name + ' param after ' + k)


@pytest.mark.parametrize('name', DCCF_CASES)
def test_dccf_eval_predict(name):
    g = load_golden(name)
    last = 's%d/after/' % (int(g['steps']) - 1)
    P = {k: g[last + k] for k in pkeys(g)}
    fw = O.dccf_forward(P, g['feat'], g['expo'], g['eval/X'], g['eval/sample_item'], g['eval/noise'], None, 0.0,
                        int(g['A']))
    close(fw['prediction'], g['eval/prediction'], FWD_RTOL, FWD_ATOL, name + ' eval')


@pytest.mark.parametrize('kind', ['RecModel', 'BiasedMF', 'IPSBiasedMF'])
def test_mf_family(kind):
    g = load_golden('mf_' + kind.lower())
    keys = [k[5:] for k in g if k.startswith('init/')]
    P = {k: g['init/' + k].copy() for k in keys}
    prop = g['propensity'] if kind == 'IPSBiasedMF' else None
    opt = O.DenseOptimizer('Adam', float(g['lr']), float(g['l2']))
    for s in range(2):
        pre = 's%d/' % s
        pred, fw = O.mf_forward(P, g[pre + 'X'], kind, prop, float(g['M']))
        close(pred, g[pre + 'prediction'], FWD_RTOL, FWD_ATOL, kind + ' pred')
        loss, dpred = O.loss_and_dpred(pred, g[pre + 'Y'], 1)
        close(loss, g[pre + 'loss'], FWD_RTOL, FWD_ATOL, kind + ' loss')
        grads = O.mf_backward(P, fw, dpred, kind)
        for k in keys:
            close(grads[k], g[pre + 'gloss/' + k], GRAD_RTOL, GRAD_ATOL, kind + ' gloss ' + k)
        P, G = O.train_step(P, opt, float(g['l2']), grads)
        for k in keys:
            close(G[k], g[pre + 'gpost/' + k], GRAD_RTOL, GRAD_ATOL, kind + ' gpost ' + k)
            # global_bias: its BPR grad is sum(dpos + dneg) == 0 up to rounding, and Adam divides by sqrt(v): the
            # update is rounding noise amplified, so only an absolute tolerance is meaningful for that scalar.
            atol = 1e-5 if k == 'global_bias' else 1e-7
            close(P[k], g[pre + 'after/' + k], 2e-5, atol, kind + ' after ' + k)
    close(O.mf_full_matrix(P, kind, prop, float(g['M'])), g['full'], FWD_RTOL, FWD_ATOL, kind + ' full')


@pytest.mark.parametrize('name', ['gd', 'adagrad', 'adam'])
def test_l2_clip_optimizers(name):
    g = load_golden('opt_' + name)
    keys = [k[5:] for k in g if k.startswith('init/')]
    P = {k: g['init/' + k].copy() for k in keys}
    opt = O.DenseOptimizer(str(g['optimizer']), float(g['lr']), float(g['l2']))
    clipped = 0
    for s in range(int(g['steps'])):
        pre = 's%d/' % s
        sparse = {k: g[pre + 'sparse/' + k] for k in keys}
        for k in keys:
            gpre = O.add_l2_grad(sparse[k], P[k], float(g['l2']))
            close(gpre, g[pre + 'gpre/' + k], 1e-6, 1e-7, 'gpre')
            clipped += int((np.abs(gpre) > 50).sum())
        P, G = O.train_step(P, opt, float(g['l2']), sparse)
        for k in keys:
            close(G[k], g[pre + 'gpost/' + k], 1e-6, 1e-7, 'gpost')
            close(P[k], g[pre + 'after/' + k], 1e-5, 1e-7, name + ' after ' + k)
    assert clipped > 0, 'the fixture must exercise the clip'


def test_batches_bit_exact():
    """Negative sampling, shuffle and feed-dict layout: integer paths, bit-exact against the reference for the same
    numpy seed and call order (main.py: test data, then train(-1), validation, then per-epoch shuffle + negatives)."""
    g = load_golden('batches')
    item_num = int(g['item_num'])
    th, vh = O.history_dicts(g['df/train'], g['df/validation'], g['df/test'])
    np.random.seed(int(g['np_seed']))
    test = O.eval_data(g['df/test'], int(g['test_neg_n']), item_num, th, vh)
    tr = g['df/train']
    train = dict(uid=tr[:, 0].copy(), iid=tr[:, 1].copy(), Y=tr[:, 2].astype(np.float32), X=tr[:, :2].copy(),
                 sample_id=np.arange(len(tr)))
    val = O.eval_data(g['df/validation'], int(g['test_neg_n']), item_num, th, vh)
    for nm, d in (('test', test), ('validation', val)):
        for k in ('uid', 'iid', 'Y', 'X', 'sample_id'):
            assert np.array_equal(d[k], g['%s/%s' % (nm, k)]), (nm, k)
    for ep in range(2):
        O.shuffle_in_unison(train)
        for k in ('uid', 'iid', 'Y', 'X', 'sample_id'):
            assert np.array_equal(train[k], g['train_ep%d/%s' % (ep, k)]), (ep, k)
        batches = O.train_batches(train, int(g['batch_size']), item_num, th)
        assert len(batches) == int(g['train_ep%d/n_batches' % ep])
        assert np.array_equal(np.concatenate([b['X'] for b in batches]), g['train_ep%d/batch_X' % ep])
        assert np.array_equal(np.concatenate([b['Y'] for b in batches]), g['train_ep%d/batch_Y' % ep])
        assert np.array_equal(np.concatenate([b['sample_id'] for b in batches]), g['train_ep%d/batch_sample_id' % ep])
        for b in batches:   # row k and row B+k carry the same uid
            B = b['real_batch_size']
            assert np.array_equal(b['X'][:B, 0], b['X'][B:, 0])
            for u, i in b['X'][B:]:
                assert int(i) not in th[int(u)]


def test_metrics():
    g = load_golden('metrics')
    vals = O.evaluate_method(g['p'], g['uid'], g['Y'], [str(m) for m in g['metrics']])
    assert np.allclose(vals, g['values'], rtol=1e-6, atol=1e-9)
    r = [3, 2, 3, 0, 0, 1, 2, 2, 3, 0]
    got = [O.dcg_at_k(r, 1, 0), O.dcg_at_k(r, 1, 1), O.dcg_at_k(r, 2, 0), O.dcg_at_k(r, 2, 1), O.dcg_at_k(r, 10, 0),
           O.dcg_at_k(r, 11, 0)]
    assert np.allclose(got, g['doc/dcg'], rtol=1e-12)
    # the docstring known answers of utils/rank_metrics.py:136-148,176-187
    assert np.allclose(got, [3.0, 3.0, 5.0, 4.2618595071429155, 9.6051177391888114, 9.6051177391888114], rtol=1e-12)
    got = [O.ndcg_at_k(r, 1, 0), O.ndcg_at_k([2, 1, 2, 0], 4, 0), O.ndcg_at_k([2, 1, 2, 0], 4, 1), O.ndcg_at_k([0], 1, 0),
           O.ndcg_at_k([1], 2, 0)]
    assert np.allclose(got, g['doc/ndcg'], rtol=1e-12)
    assert np.allclose(got, [1.0, 0.9203032077642922, 0.96519546960144276, 0.0, 1.0], rtol=1e-12)


def test_eval_negatives_oracle_properties():
    """oracle/philox.py::eval_negatives (the sequential definition the device sampler is checked against): neg_n distinct
    admissible items per user in draw order, reproducible, split-dependent, -1 rows where too few items remain."""
    from oracle import philox as PH
    rng = np.random.RandomState(1)
    U_, I_, neg_n = 40, 97, 15
    hist = [np.sort(rng.choice(I_, rng.randint(0, 30), replace=False)) for _ in range(U_)]
    hist[3] = np.sort(rng.choice(I_, 85, replace=False))          # 12 items left < neg_n -> -1
    hist[4] = np.sort(rng.choice(np.arange(1, I_), 80, replace=False))          # low regime: 17 left incl. item 0
    indptr = np.r_[0, np.cumsum([len(h) for h in hist])].astype(np.int64)
    items = np.concatenate(hist).astype(np.int64)
    users = np.arange(U_, dtype=np.int64)
    a = PH.eval_negatives(5, 1, users, I_, indptr, items, neg_n)
    assert np.array_equal(a, PH.eval_negatives(5, 1, users, I_, indptr, items, neg_n))
    assert not np.array_equal(a, PH.eval_negatives(5, 2, users, I_, indptr, items, neg_n))
    assert (a[3] == -1).all()
    for u in range(U_):
        if u == 3:
            continue
        assert len(set(a[u].tolist())) == neg_n and not set(a[u].tolist()) & set(hist[u].tolist())
    assert 0 not in a[4]                                           # item 0 is never drawn once < 20 % of the items remain
    # draw order: the first accepted draw of user 0 is its first admissible Philox candidate
    k0, k1 = PH._key(5, PH.STREAM_EVALNEG)
    xs = PH.philox4x32(0, np.arange(8), 1, 0, k0, k1)
    cand = np.stack([PH.mulhi(x, I_) for x in xs], axis=1).reshape(-1)
    first = next(int(c) for c in cand if int(c) not in set(hist[0].tolist()))
    assert a[0][0] == first


def test_reference_checkpoint_fixture_matches_golden_parameters():
    """tests/golden/dccf_d24_f100_l3_adagrad.pt was written by the reference's own BaseModel.save_model (src/models/BaseModel.py:
    224-236) after the golden's training steps: its keys, order, shapes and values are the golden's last `after/` parameters."""
    import os
    import torch
    from conftest import GOLDEN
    g = load_golden('dccf_d24_f100_l3_adagrad')
    sd = torch.load(os.path.join(GOLDEN, 'dccf_d24_f100_l3_adagrad.pt'), map_location='cpu')
    keys = pkeys(g)
    assert list(sd.keys()) == keys == ['uid_embeddings.weight', 'iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias',
                                       'mlp.1.weight', 'mlp.1.bias', 'mlp.2.weight', 'mlp.2.bias']
    last = 's%d/after/' % (int(g['steps']) - 1)
    for k in keys:
        assert np.array_equal(sd[k].numpy(), g[last + k])


def test_epoch_permutation_is_a_bijection_and_uniform():
    """The keyed bijection that replaces torch.randperm for the epoch's in-unison shuffle (k_epoch_batches; ADVICE r2: it had
    coverage tests only — and round 2's multiply-add-xorshift rounds fail THIS test at n = 7 with p ~ 1e-43).  For every
    (seed, epoch, n): a permutation.  Over many epochs: position -> value counts are uniform (chi-square per position and over
    all positions), and so are the differences of neighbouring values (no visible structure in (perm[i], perm[i + 1]))."""
    from oracle import philox as PH
    from scipy.stats import chi2
    for n in (1, 2, 3, 7, 130, 1000, 1025):
        for seed, epoch in ((2019, 0), (2019, 1), (7, 123456)):
            p = PH.epoch_perm(seed, epoch, n)
            assert sorted(p.tolist()) == list(range(n)), (n, seed, epoch)
    assert not np.array_equal(PH.epoch_perm(2019, 0, 130), PH.epoch_perm(2019, 1, 130))
    assert not np.array_equal(PH.epoch_perm(2019, 0, 130), PH.epoch_perm(2020, 0, 130))
    for n, E in ((5, 5000), (7, 7000), (130, 6500)):
        C = np.zeros((n, n))
        Dm = np.zeros(n)
        for e in range(E):
            p = PH.epoch_perm(2019, e, n)
            C[np.arange(n), p] += 1
            Dm += np.bincount((p[1:] - p[:-1]) % n, minlength=n)
        exp = E / n
        stat = ((C - exp) ** 2 / exp).sum(1)                    # one chi-square (n - 1 dof) per position
        pv = chi2.sf(stat, n - 1)
        assert pv.min() > 1e-4 / n, (n, pv.min())               # (Bonferroni over the n positions)
        # all positions together: the sum of n chi-squares (n (n - 1) dof; the positions are only mildly dependent)
        assert chi2.sf(stat.sum(), n * (n - 1)) > 1e-4, (n, stat.sum())
        # neighbour differences: 0 is impossible (a permutation), the other n - 1 residues are equally likely
        d = Dm[1:]
        sd = ((d - d.mean()) ** 2 / d.mean()).sum()
        assert chi2.sf(sd, n - 2) > 1e-4, (n, sd)
