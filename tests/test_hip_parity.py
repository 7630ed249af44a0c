# coding=utf-8
"""Parity of the HIP path (through the C ABI) against the golden vectors of the reference and against the oracle.
Run on the GPU box:  python -m pytest tests -m gpu"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dccf_oracle as O
from oracle import philox as PH

pytestmark = pytest.mark.gpu

DCCF_CASES = ['dccf_d16_f32_adam', 'dccf_d64_f768_adam', 'dccf_d64_f32_nodrop_gd', 'dccf_d32_f160_adagrad',
              'dccf_d128_f768_adam', 'dccf_d64_f768_mse',
              # round 2: --n_layers 2 / 3 (src/models/DCCF.py:61-62,91-94), widths that are not a kernel tile
              # (src/models/RecModel.py:17-27), feature files wider than 896 (src/models/DCCF.py:59)
              'dccf_d64_f768_l2_adam', 'dccf_d24_f100_l3_adagrad', 'dccf_d48_f1024_adam', 'dccf_d100_f800_gd',
              'dccf_d128_f32_l2_mse',
              # round 3: embedding sizes above 128 (the 256 column tile with a run-time width; src/models/RecModel.py:17-27)
              'dccf_d192_f768_adam', 'dccf_d256_f96_adagrad', 'dccf_d160_f1000_gd_mse']
PKEYS = ['uid_embeddings.weight', 'iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias']


def pkeys(g):
    """state_dict keys of a golden case in the reference's order: PKEYS + mlp.k.weight / mlp.k.bias for k >= 1."""
    return [k[5:] for k in g if k.startswith('init/')]


def extra_of(views, keys):
    """[(mlp.k.weight, mlp.k.bias) for k >= 1] out of a dict of tensors."""
    n = (len(keys) - 4) // 2
    return [(views['mlp.%d.weight' % k], views['mlp.%d.bias' % k]) for k in range(1, n + 1)]

# fp32 tolerances (relative to the largest magnitude of the compared tensor): summation order differs between
# ATen / numpy / the MFMA k-order, and float atomics reorder the scatter sums.
FWD_RTOL, FWD_ATOL = 3e-5, 1e-6
GRAD_RTOL, GRAD_ATOL = 1e-4, 2e-6
PARAM_RTOL, PARAM_ATOL = 3e-5, 2e-7
# After an optimizer step: Adam/Adagrad move every element by ~lr * g/|g|, which turns the RELATIVE error of a small
# gradient element into an absolute parameter error of lr * rel_err; agreement is therefore stated in fractions of one
# step: 0.5 % of lr on top of the fp32 tolerance above.
STEP_FRAC = 5e-3


def dev():
    return torch.device('cuda:0')


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev()).contiguous()


def close(a, b, rtol, atol, what=''):
    a = a.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * scale, '%s: max err %g (scale %g)' % (what, err, scale)


@pytest.fixture(scope='module')
def L():
    from dccf_amd import _lib
    _lib.load()
    return _lib


@pytest.fixture(scope='module')
def ctx(L):
    return L.Context(0)


class FlatParams(object):
    """One flat fp32 buffer [U | V | W | b] with views, plus grads and optimizer state — how the host side holds them."""

    def __init__(self, P):
        self.keys = list(P.keys())
        sizes = [int(np.prod(P[k].shape)) for k in self.keys]
        pad = [(s + 3) // 4 * 4 for s in sizes]
        n = sum(pad)
        self.p = torch.zeros(n, dtype=torch.float32, device=dev())
        self.g = torch.zeros_like(self.p)
        self.s1 = torch.zeros_like(self.p)
        self.s2 = torch.zeros_like(self.p)
        self.views, self.gviews = {}, {}
        o = 0
        for k, s, pd in zip(self.keys, sizes, pad):
            self.views[k] = self.p[o:o + s].view(*P[k].shape)
            self.gviews[k] = self.g[o:o + s].view(*P[k].shape)
            self.views[k].copy_(T(P[k]))
            o += pd


@pytest.mark.parametrize('name', DCCF_CASES)
def test_dccf_injected_train_steps_match_reference(L, ctx, name):
    g = load_golden(name)
    A, S, p, rank = int(g['A']), int(g['S']), float(g['dropout']), int(g['rank'])
    keys = pkeys(g)
    fp = FlatParams({k: g['init/' + k] for k in keys})
    feat, expo = T(g['feat']), T(g['expo'])
    kind = str(g['optimizer']).lower()
    for s in range(int(g['steps'])):
        pre = 's%d/' % s
        v = fp.views
        m = L.model_struct(v[PKEYS[0]], v[PKEYS[1]], v[PKEYS[2]], v[PKEYS[3]], feat, expo, S, A, float(g['std']),
                           extra=extra_of(v, keys))
        X, Y = T(g[pre + 'X']), T(g[pre + 'Y'])
        r = L.rand_struct(sample_item=T(g[pre + 'sample_item']), noise=T(g[pre + 'noise']),
                          keep=T(g[pre + 'mask']) if p > 0 else None)
        # predict-only entry point on the same draws
        close(L.dccf_predict(ctx, m, r, X, p), g[pre + 'prediction'], FWD_RTOL, FWD_ATOL, name + ' predict')
        gv = fp.gviews
        pred, loss = L.dccf_train_fwdbwd(ctx, m, r, X, Y, rank, p, gv[PKEYS[0]], gv[PKEYS[1]], gv[PKEYS[2]], gv[PKEYS[3]],
                                         gextra=extra_of(gv, keys))
        close(pred, g[pre + 'prediction'], FWD_RTOL, FWD_ATOL, name + ' pred')
        close(loss, g[pre + 'loss'].reshape(1), FWD_RTOL, FWD_ATOL, name + ' loss')
        for k in keys:
            close(gv[k], g[pre + 'gloss/' + k], GRAD_RTOL, GRAD_ATOL, name + ' gloss ' + k)
        close(L.sumsq(fp.p), g[pre + 'l2'].reshape(1), 1e-5, 0, name + ' l2')
        L.dense_opt_step(kind, fp.p, fp.g, fp.s1, fp.s2, float(g['lr']), float(g['l2']), float(g['l2']), 50.0, s + 1)
        assert float(fp.g.abs().max()) == 0.0     # fused zero_grad
        for k in keys:
            close(fp.views[k], g[pre + 'after/' + k], PARAM_RTOL, PARAM_ATOL + STEP_FRAC * float(g['lr']), name + ' after ' + k)


@pytest.mark.parametrize('lazy_K', [0, 2, 8])
@pytest.mark.parametrize('name', DCCF_CASES)
def test_dccf_golden_cases_through_the_default_train_step(L, name, lazy_K):
    """The reference's golden cases through what runner.fit and bench.py actually run: DCCF.train_step = ONE dccf_train_step call
    per batch under the windowed lazy regularisation (lazy_K = 2: every window comes round inside the case's 2-3 steps; 8: the
    default; 0: the dense pass), with the reference's captured draws injected.  Parameters after every step against `after/*`
    (state_dict() flushes the rows that are behind), the loss and the prediction against the reference's."""
    from dccf_amd.models import DCCF, FusedOptimizer
    g = load_golden(name)
    A, S, p, rank = int(g['A']), int(g['S']), float(g['dropout']), int(g['rank'])
    keys = pkeys(g)
    U, D = g['init/' + PKEYS[0]].shape
    I = g['init/' + PKEYS[1]].shape[0]
    m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=S, attribute_num=A, std=float(g['std']), label_min=0,
             label_max=1, feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=(len(keys) - 2) // 2,
             random_seed=1, model_path='/tmp/x.pt', feature_embedding=T(g['feat']), expo_prob=T(g['expo']))
    m.load_state_dict({k: T(g['init/' + k]) for k in keys})
    m.optimizer = FusedOptimizer(m, str(g['optimizer']).lower(), float(g['lr']), float(g['l2']))
    m.lazy_K = lazy_K
    m.train()
    for s in range(int(g['steps'])):
        pre = 's%d/' % s
        batch = {'X': T(g[pre + 'X']), 'Y': T(g[pre + 'Y']), 'rank': rank, 'train': True, 'dropout': p,
                 'inject': {'sample_item': T(g[pre + 'sample_item']), 'noise': T(g[pre + 'noise']),
                            'keep': T(g[pre + 'mask']) if p > 0 else None}}
        out = m.train_step(batch)
        assert (m.optimizer.lazy is not None) == (lazy_K >= 2 and D % 4 == 0)        # (rows of whole float4 slots: D = 24 / 48 / 100 too)
        close(out['prediction'], g[pre + 'prediction'], FWD_RTOL, FWD_ATOL, name + ' pred')
        close(out['loss'].reshape(1), g[pre + 'loss'].reshape(1), FWD_RTOL, FWD_ATOL, name + ' loss')
        sd = m.state_dict()
        for k in keys:
            close(sd[k], g[pre + 'after/' + k], PARAM_RTOL, PARAM_ATOL + STEP_FRAC * float(g['lr']), name + ' after ' + k)
        assert float(m.flat_g.abs().max()) == 0.0
        m.train()


@pytest.mark.parametrize('name', DCCF_CASES)
def test_dccf_injected_eval_predict(L, ctx, name):
    g = load_golden(name)
    last = 's%d/after/' % (int(g['steps']) - 1)
    keys = pkeys(g)
    Pd = {k: T(g[last + k]) for k in keys}
    P = [Pd[k] for k in PKEYS]
    m = L.model_struct(P[0], P[1], P[2], P[3], T(g['feat']), T(g['expo']), int(g['S']), int(g['A']), float(g['std']),
                       extra=extra_of(Pd, keys))
    r = L.rand_struct(sample_item=T(g['eval/sample_item']), noise=T(g['eval/noise']))
    close(L.dccf_predict(ctx, m, r, T(g['eval/X']), 0.0), g['eval/prediction'], FWD_RTOL, FWD_ATOL, name + ' eval')


def test_fused_streams_match_oracle(L):
    """The on-device random streams against oracle/philox.py: integers bit-exact, normals to fp32 rounding."""
    seed, step = 0x1234567890ABCDEF, 77
    c = L.debug_candidates(37, 10, 5000, seed, step, dev()).cpu().numpy()
    assert np.array_equal(c, PH.candidates(seed, step, 37, 10, 5000))
    assert c.min() >= 0 and c.max() < 5000
    for p in (0.2, 0.5):
        k = L.debug_keep(201, 64, p, seed, step, dev()).cpu().numpy()
        assert np.array_equal(k, PH.dropout_keep(seed, step, 201, 64, float(np.float32(p))))
    assert L.debug_keep(50, 64, 0.0, seed, step, dev()).cpu().numpy().min() == 1
    for F in (32, 160, 768):
        z = L.debug_noise(97, F, 0.1, seed, step, dev()).cpu().numpy()
        zo = PH.noise(seed, step, 97, F, 0.1)
        assert np.abs(z - zo).max() <= 2e-6 * 0.1 * 6 + 1e-7, np.abs(z - zo).max()
    # distribution sanity of the device normals
    z = L.debug_noise(4096, 768, 1.0, 99, 3, dev()).double()
    assert abs(float(z.mean())) < 2e-3 and abs(float(z.std()) - 1.0) < 2e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 3e-2


@pytest.mark.parametrize('D,F,S,A,p,rank', [(64, 768, 10, 2, 0.2, 1), (16, 32, 10, 2, 0.2, 1), (128, 768, 10, 2, 0.2, 1),
                                            (32, 160, 4, 3, 0.5, 0), (64, 768, 10, 2, 0.0, 1),
                                            (64, 160, 20, 1, 0.2, 1), (32, 32, 3, 5, 0.1, 1), (16, 96, 40, 2, 0.3, 0),
                                            # config 1's shape (D=16, F=768) and the chunk counts / column clamps no golden
                                            # reaches: 6 whole chunks at D=16/32, 3 chunks padded to 6, 7 chunks with a clamp
                                            (16, 768, 10, 2, 0.2, 1), (32, 768, 3, 2, 0.2, 1), (64, 384, 3, 2, 0.2, 1),
                                            (64, 800, 3, 2, 0.1, 1), (128, 300, 2, 2, 0.2, 0)])
def test_fused_equals_injected_on_device_draws(L, ctx, D, F, S, A, p, rank):
    """Fused mode (Philox in registers) must equal injected mode fed with the very same draws written out by the
    debug entry points — and both must equal the oracle on those draws."""
    rng = np.random.RandomState(D + F)
    U_, I_, pairs = 300, 200, 37
    N = 2 * pairs
    P = {'uid_embeddings.weight': (rng.randn(U_, D) * 0.3).astype(np.float32),
         'iid_embeddings.weight': (rng.randn(I_, D) * 0.3).astype(np.float32),
         'mlp.0.weight': (rng.randn(D, D + F) * 0.1).astype(np.float32),
         'mlp.0.bias': (rng.randn(D) * 0.1).astype(np.float32)}
    feat = (rng.randn(I_, F) * 0.5).astype(np.float32)
    expo = rng.randn(U_, I_).astype(np.float32)
    u = rng.randint(0, U_, pairs)
    X = np.concatenate([np.stack([u, rng.randint(0, I_, pairs)], 1), np.stack([u, rng.randint(0, I_, pairs)], 1)]).astype(np.int64)
    Y = rng.randint(0, 2, N).astype(np.float32)
    seed, step, std = 20191104, 5, 0.1
    Ld = N * (S + 1) * A
    tp = {k: T(v) for k, v in P.items()}
    m = L.model_struct(tp[PKEYS[0]], tp[PKEYS[1]], tp[PKEYS[2]], tp[PKEYS[3]], T(feat), T(expo), S, A, std)
    si = L.debug_candidates(N, S, I_, seed, step, dev())
    nz = L.debug_noise(Ld, F, std, seed, step, dev())
    kp = L.debug_keep(Ld, D, p, seed, step, dev())
    outs = []
    for r in (L.rand_struct(seed=seed, step=step), L.rand_struct(sample_item=si, noise=nz, keep=kp if p > 0 else None)):
        gr = [torch.zeros_like(tp[k]) for k in PKEYS]
        pred, loss = L.dccf_train_fwdbwd(ctx, m, r, T(X), T(Y), rank, p, *gr)
        torch.cuda.synchronize()
        outs.append((pred.cpu().numpy(), loss.cpu().numpy(), [x.cpu().numpy() for x in gr]))
    # same arithmetic, same order -> forward bit-exact; grads equal up to atomic ordering
    assert np.array_equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][2], outs[1][2]):
        close(a, b, 2e-6, 1e-8, 'fused vs injected grads')
    fw = O.dccf_forward(P, feat, expo, X, si.cpu().numpy(), nz.cpu().numpy(), kp.cpu().numpy(), p, A)
    close(outs[0][0], fw['prediction'], FWD_RTOL, FWD_ATOL, 'fused pred vs oracle')
    loss, dpred = O.loss_and_dpred(fw['prediction'], Y, rank)
    close(outs[0][1], np.asarray(loss).reshape(1), FWD_RTOL, FWD_ATOL, 'fused loss vs oracle')
    grads = O.dccf_backward(P, fw, dpred, A)
    for k, a in zip(PKEYS, outs[0][2]):
        close(a, grads[k], GRAD_RTOL, GRAD_ATOL, 'fused grad vs oracle ' + k)


@pytest.mark.parametrize('D,F,S,A,p,rank,NL', [(64, 768, 10, 2, 0.2, 1, 2), (24, 100, 4, 2, 0.3, 1, 3), (128, 160, 3, 2, 0.2, 0, 2),
                                               (16, 32, 10, 2, 0.5, 1, 4), (48, 1024, 3, 2, 0.2, 1, 1), (100, 1500, 2, 2, 0.1, 1, 2),
                                               (7, 33, 5, 3, 0.2, 0, 1)])
def test_fused_equals_injected_extra_layers_generic_shapes(L, ctx, D, F, S, A, p, rank, NL):
    """--n_layers > 1 (src/models/DCCF.py:61-62,91-94), any embedding width (src/models/RecModel.py:17-27) and any feature
    width (src/models/DCCF.py:59): fused draws == injected draws (the per-layer keep masks written out by
    dccf_debug_keep_layer) == the oracle on those draws."""
    rng = np.random.RandomState(1000 * NL + D + F)
    U_, I_, pairs = 120, 90, 19
    N = 2 * pairs
    keys = list(PKEYS)
    P = {'uid_embeddings.weight': (rng.randn(U_, D) * 0.3).astype(np.float32),
         'iid_embeddings.weight': (rng.randn(I_, D) * 0.3).astype(np.float32),
         'mlp.0.weight': (rng.randn(D, D + F) * 0.1).astype(np.float32),
         'mlp.0.bias': (rng.randn(D) * 0.1).astype(np.float32)}
    for k in range(1, NL):
        P['mlp.%d.weight' % k] = (rng.randn(D, D) * (0.8 / np.sqrt(D))).astype(np.float32)
        P['mlp.%d.bias' % k] = (rng.randn(D) * 0.1).astype(np.float32)
        keys += ['mlp.%d.weight' % k, 'mlp.%d.bias' % k]
    feat = (rng.randn(I_, F) * 0.5).astype(np.float32)
    expo = rng.randn(U_, I_).astype(np.float32)
    u = rng.randint(0, U_, pairs)
    X = np.concatenate([np.stack([u, rng.randint(0, I_, pairs)], 1), np.stack([u, rng.randint(0, I_, pairs)], 1)]).astype(np.int64)
    Y = rng.randint(0, 2, N).astype(np.float32)
    seed, step, std = 77001, 9, 0.1
    Ld = N * (S + 1) * A
    tp = {k: T(v) for k, v in P.items()}
    m = L.model_struct(tp[PKEYS[0]], tp[PKEYS[1]], tp[PKEYS[2]], tp[PKEYS[3]], T(feat), T(expo), S, A, std, extra=extra_of(tp, keys))
    si = L.debug_candidates(N, S, I_, seed, step, dev())
    nz = L.debug_noise(Ld, F, std, seed, step, dev())
    kp = torch.stack([L.debug_keep(Ld, D, p, seed, step, dev(), layer=k) for k in range(NL)]).contiguous()
    kpo = np.stack([PH.dropout_keep(seed, step, Ld, D, float(np.float32(p)), layer=k) for k in range(NL)])
    assert np.array_equal(kp.cpu().numpy(), kpo)
    assert NL == 1 or not np.array_equal(kpo[0], kpo[1])
    outs = []
    for r in (L.rand_struct(seed=seed, step=step), L.rand_struct(sample_item=si, noise=nz, keep=kp)):
        gr = {k: torch.zeros_like(tp[k]) for k in keys}
        pred, loss = L.dccf_train_fwdbwd(ctx, m, r, T(X), T(Y), rank, p, *[gr[k] for k in PKEYS], gextra=extra_of(gr, keys))
        pe = L.dccf_predict(ctx, m, r, T(X), p)
        torch.cuda.synchronize()
        assert torch.equal(pe, pred)           # the predict-only entry runs the same forward
        outs.append((pred.cpu().numpy(), loss.cpu().numpy(), {k: v.cpu().numpy() for k, v in gr.items()}))
    assert np.array_equal(outs[0][0], outs[1][0])
    for k in keys:
        close(outs[0][2][k], outs[1][2][k], 2e-6, 1e-8, 'fused vs injected grads ' + k)
    fw = O.dccf_forward(P, feat, expo, X, si.cpu().numpy(), nz.cpu().numpy(), kpo, p, A)
    close(outs[0][0], fw['prediction'], FWD_RTOL, FWD_ATOL, 'fused pred vs oracle')
    loss, dpred = O.loss_and_dpred(fw['prediction'], Y, rank)
    close(outs[0][1], np.asarray(loss).reshape(1), FWD_RTOL, FWD_ATOL, 'fused loss vs oracle')
    grads = O.dccf_backward(P, fw, dpred, A)
    for k in keys:
        close(outs[0][2][k], grads[k], GRAD_RTOL, GRAD_ATOL, 'fused grad vs oracle ' + k)


def test_dccf_on_the_fly_exposure_equals_dense(L, ctx):
    """expo == NULL: exposure computed from the IPSBiasedMF factors must equal the dense-matrix path fed with
    mf_predict_full of the same factors (README.md:28-30)."""
    rng = np.random.RandomState(3)
    U_, I_, D, F, S, A = 150, 260, 64, 768, 10, 2
    ips = dict(P=T((rng.randn(U_, 32) * 0.3).astype(np.float32)), Q=T((rng.randn(I_, 32) * 0.3).astype(np.float32)),
               bu=T((rng.randn(U_) * 0.1).astype(np.float32)), bi=T((rng.randn(I_) * 0.1).astype(np.float32)),
               prop=T(rng.rand(I_).astype(np.float32)), b0=0.1, M=0.1)
    b0 = torch.full((1,), 0.1, device=dev())
    mf = L.mf_struct('IPSBiasedMF', ips['P'], ips['Q'], ips['bu'], ips['bi'], b0, ips['prop'], 0.1)
    dense = L.mf_predict_full(mf, device=dev())
    Pm = [T((rng.randn(U_, D) * 0.3).astype(np.float32)), T((rng.randn(I_, D) * 0.3).astype(np.float32)),
          T((rng.randn(D, D + F) * 0.1).astype(np.float32)), T((rng.randn(D) * 0.1).astype(np.float32))]
    feat = T((rng.randn(I_, F) * 0.5).astype(np.float32))
    X = T(np.stack([rng.randint(0, U_, 64), rng.randint(0, I_, 64)], 1).astype(np.int64))
    r = L.rand_struct(seed=11, step=1)
    a = L.dccf_predict(ctx, L.model_struct(Pm[0], Pm[1], Pm[2], Pm[3], feat, dense, S, A, 0.1), r, X, 0.0)
    b = L.dccf_predict(ctx, L.model_struct(Pm[0], Pm[1], Pm[2], Pm[3], feat, None, S, A, 0.1, ips=ips), r, X, 0.0)
    close(a, b.cpu().numpy(), 2e-5, 1e-6, 'on-the-fly exposure')


@pytest.mark.parametrize('kind', ['RecModel', 'BiasedMF', 'IPSBiasedMF'])
def test_mf_family_matches_reference(L, ctx, kind):
    g = load_golden('mf_' + kind.lower())
    keys = [k[5:] for k in g if k.startswith('init/')]
    fp = FlatParams({k: g['init/' + k].reshape(-1) if g['init/' + k].ndim == 0 else g['init/' + k] for k in keys})
    prop = T(g['propensity']) if kind == 'IPSBiasedMF' else None

    def struct():
        v = fp.views
        if kind == 'RecModel':
            return L.mf_struct(kind, v['uid_embeddings.weight'], v['iid_embeddings.weight'])
        return L.mf_struct(kind, v['uid_embeddings.weight'], v['iid_embeddings.weight'], v['user_bias.weight'].view(-1),
                           v['item_bias.weight'].view(-1), v['global_bias'].view(-1), prop, float(g['M']))

    for s in range(2):
        pre = 's%d/' % s
        X, Y = T(g[pre + 'X']), T(g[pre + 'Y'])
        m = struct()
        close(L.mf_predict(m, X), g[pre + 'prediction'], FWD_RTOL, FWD_ATOL, kind + ' predict')
        gv = fp.gviews
        if kind == 'RecModel':
            pred, loss = L.mf_train_fwdbwd(ctx, m, X, Y, 1, gv['uid_embeddings.weight'], gv['iid_embeddings.weight'])
        else:
            pred, loss = L.mf_train_fwdbwd(ctx, m, X, Y, 1, gv['uid_embeddings.weight'], gv['iid_embeddings.weight'],
                                           gv['user_bias.weight'].view(-1), gv['item_bias.weight'].view(-1),
                                           gv['global_bias'].view(-1))
        close(pred, g[pre + 'prediction'], FWD_RTOL, FWD_ATOL, kind + ' pred')
        close(loss, g[pre + 'loss'].reshape(1), FWD_RTOL, FWD_ATOL, kind + ' loss')
        for k in keys:
            close(gv[k].view(g[pre + 'gloss/' + k].shape), g[pre + 'gloss/' + k], GRAD_RTOL, GRAD_ATOL, kind + ' gloss ' + k)
        L.dense_opt_step('adam', fp.p, fp.g, fp.s1, fp.s2, float(g['lr']), float(g['l2']), float(g['l2']), 50.0, s + 1)
        for k in keys:
            atol = 1e-5 if k == 'global_bias' else PARAM_ATOL + STEP_FRAC * float(g['lr'])   # see tests/test_oracle_golden.py
            close(fp.views[k].view(g[pre + 'after/' + k].shape), g[pre + 'after/' + k], PARAM_RTOL, atol, kind + ' after ' + k)
    close(L.mf_predict_full(struct(), device=dev()), g['full'], FWD_RTOL, FWD_ATOL, kind + ' full matrix')


@pytest.mark.parametrize('form', ['lines', 'rows'])
@pytest.mark.parametrize('D', [16, 32, 64, 128, 24, 100, 7, 1, 160, 255, 256])
@pytest.mark.parametrize('kind', ['RecModel', 'BiasedMF', 'IPSBiasedMF'])
def test_mf_full_matrix_every_kernel_form_ragged_sizes(L, D, kind, form, monkeypatch):
    """mf_predict_full for every D-specific kernel form (the line form of round 3 and, with DCCF_FULL_FORM=rows, round 2's row-band
    form for D = 16 / 32 / 64 / 128; the generic tile for 24) at sizes that are multiples of nothing (user bands, 32-, 64- and 256-item
    tiles all end ragged) against the formula in fp64 (src/models/IPSBiasedMF.py:37-57: (P Q^T + bu + bi + b0) / max(prop, M))."""
    if form == 'rows':
        if D > 128 or D % 2:
            pytest.skip('the older forms cover even sizes up to 128')
        monkeypatch.setenv('DCCF_FULL_FORM', 'rows')
    else:
        monkeypatch.delenv('DCCF_FULL_FORM', raising=False)
    U, I = 257 + 31, 2 * 256 + 77
    g = torch.Generator(device='cuda').manual_seed(D)
    P, Q = torch.randn(U, D, generator=g, device='cuda') * 0.3, torch.randn(I, D, generator=g, device='cuda') * 0.3
    bu, bi = torch.randn(U, generator=g, device='cuda') * 0.1, torch.randn(I, generator=g, device='cuda') * 0.1
    prop = torch.rand(I, generator=g, device='cuda')
    b0 = torch.full((1,), 0.1, device='cuda')
    m = L.mf_struct(kind, P, Q, bu, bi, b0, prop, 0.3)
    out = torch.full((U + 1, I), float('nan'), device='cuda')          # one guard row: nothing may be written past the end
    L.mf_predict_full(m, out=out[:U])
    ref = P.double() @ Q.double().T
    if kind != 'RecModel':
        ref = ref + bu.double()[:, None] + bi.double()[None, :] + 0.1
    if kind == 'IPSBiasedMF':
        ref = ref / torch.clamp(prop.double(), min=0.3)[None, :]
    assert bool(torch.isfinite(out[:U]).all()) and bool(torch.isnan(out[U]).all())
    close(out[:U], ref.cpu().numpy(), 2e-6, 1e-6 if D <= 128 else 2e-6, 'full matrix D=%d %s' % (D, kind))


@pytest.mark.parametrize('U,I,off', [(1, 1, 0), (5, 31, 3), (33, 32, 1), (130, 64, 7), (31, 257, 31), (64, 1000, 13), (257, 4101, 2),
                                     (128, 96, 0), (129, 33, 5)])
@pytest.mark.parametrize('D', [16, 64, 128])
def test_mf_full_matrix_line_form_edges(L, D, U, I, off, monkeypatch):
    """The line form's stores are aligned to the 128-byte lines of `out` whatever the row length and the pointer are: tiny matrices
    (fewer item tiles than item ranges, fewer users than a band), row lengths that ARE multiples of 32, and an `out` that starts
    `off` floats into its allocation.  Nothing outside [out, out + U * I) may be written, everything inside must be
    (README.md:28-30; formula src/models/IPSBiasedMF.py:37-57)."""
    monkeypatch.delenv('DCCF_FULL_FORM', raising=False)
    g = torch.Generator(device='cuda').manual_seed(D + U + I)
    P, Q = torch.randn(U, D, generator=g, device='cuda') * 0.3, torch.randn(I, D, generator=g, device='cuda') * 0.3
    bu, bi = torch.randn(U, generator=g, device='cuda') * 0.1, torch.randn(I, generator=g, device='cuda') * 0.1
    prop = torch.rand(I, generator=g, device='cuda')
    b0 = torch.full((1,), 0.1, device='cuda')
    m = L.mf_struct('IPSBiasedMF', P, Q, bu, bi, b0, prop, 0.3)
    flat = torch.full((off + U * I + 200,), float('nan'), device='cuda')
    out = flat[off:off + U * I].view(U, I)
    L.mf_predict_full(m, out=out)
    ref = (P.double() @ Q.double().T + bu.double()[:, None] + bi.double()[None, :] + 0.1) / torch.clamp(prop.double(), min=0.3)[None, :]
    assert bool(torch.isnan(flat[:off]).all()) and bool(torch.isnan(flat[off + U * I:]).all()), 'wrote outside the matrix'
    assert bool(torch.isfinite(out).all()), 'left a hole'
    close(out, ref.cpu().numpy(), 2e-6, 1e-6, 'full matrix D=%d %dx%d +%d' % (D, U, I, off))


@pytest.mark.parametrize('D', [64, 128])
def test_mf_full_matrix_at_config4_size_properties(L, D):
    """BASELINE config 4 at its full size (CDs-and-Vinyl-shaped 75,258 x 64,443 = 19.4 GB; README.md:28-30), where no dense reference
    fits a test: (1) two million random entries + the matrix's four corners equal the PAIRWISE predict of the same model
    (mf_predict: src/models/IPSBiasedMF.py:37-57 on (u, i) rows) to fp32 rounding of two summation orders; (2) linearity — for the plain
    dot product the row sums equal P (sum_i Q_i) and the column sums (sum_u P_u) Q^T, accumulated in fp64; (3) the guard floats before
    and after the matrix are untouched."""
    U, I = 75258, 64443
    g = torch.Generator(device='cuda').manual_seed(D)
    P, Q = torch.randn(U, D, generator=g, device='cuda') * 0.1, torch.randn(I, D, generator=g, device='cuda') * 0.1
    bu, bi = torch.randn(U, generator=g, device='cuda') * 0.1, torch.randn(I, generator=g, device='cuda') * 0.1
    prop = torch.rand(I, generator=g, device='cuda')
    b0 = torch.full((1,), 0.1, device='cuda')
    flat = torch.full((64 + U * I + 64,), float('nan'), device='cuda')
    out = flat[64:64 + U * I].view(U, I)
    m = L.mf_struct('IPSBiasedMF', P, Q, bu, bi, b0, prop, 0.1)
    L.mf_predict_full(m, out=out)
    X = torch.stack([torch.randint(0, U, (2000000,), generator=g, device='cuda'), torch.randint(0, I, (2000000,), generator=g, device='cuda')], 1)
    X[:4] = torch.tensor([[0, 0], [0, I - 1], [U - 1, 0], [U - 1, I - 1]], device='cuda')
    pair = L.mf_predict(m, X.contiguous())
    got = out[X[:, 0], X[:, 1]]
    assert bool(torch.isfinite(got).all())
    err = (got - pair).abs() / (1.0 + pair.abs())
    assert float(err.max()) < 3e-6, float(err.max())
    assert bool(torch.isnan(flat[:64]).all()) and bool(torch.isnan(flat[-64:]).all())
    # linearity of the bare product
    m0 = L.mf_struct('RecModel', P, Q, bu, bi, b0, prop, 0.1)
    L.mf_predict_full(m0, out=out)
    rows = out.sum(1, dtype=torch.float64)
    cols = out.sum(0, dtype=torch.float64)
    ref_r = P.double() @ Q.double().sum(0)
    ref_c = Q.double() @ P.double().sum(0)
    assert float((rows - ref_r).abs().max()) < 2e-3 and float((cols - ref_c).abs().max()) < 2e-3      # sums of 6e4 .. 8e4 fp32 values
    assert bool(torch.isnan(flat[:64]).all()) and bool(torch.isnan(flat[-64:]).all())


def test_mf_train_mse_and_duplicates_vs_oracle(L, ctx):
    """rank==0 (MSE) and a batch full of duplicate users/items (exercises the LDS duplicate-row chains)."""
    rng = np.random.RandomState(8)
    U_, I_, D, N = 9, 7, 64, 333
    P = {'uid_embeddings.weight': rng.randn(U_, D).astype(np.float32) * 0.3,
         'iid_embeddings.weight': rng.randn(I_, D).astype(np.float32) * 0.3,
         'user_bias.weight': rng.randn(U_, 1).astype(np.float32) * 0.1,
         'item_bias.weight': rng.randn(I_, 1).astype(np.float32) * 0.1,
         'global_bias': np.float32(0.1)}
    prop = rng.rand(I_).astype(np.float32)
    X = np.stack([rng.randint(0, U_, N), rng.randint(0, I_, N)], 1).astype(np.int64)
    Y = rng.randint(0, 2, N).astype(np.float32)
    pred_o, fw = O.mf_forward(P, X, 'IPSBiasedMF', prop, 0.1)
    loss_o, dpred = O.loss_and_dpred(pred_o, Y, 0)
    go = O.mf_backward(P, fw, dpred, 'IPSBiasedMF')
    t = {k: T(np.asarray(v).reshape(-1) if np.ndim(v) == 0 else v) for k, v in P.items()}
    m = L.mf_struct('IPSBiasedMF', t['uid_embeddings.weight'], t['iid_embeddings.weight'], t['user_bias.weight'].view(-1),
                    t['item_bias.weight'].view(-1), t['global_bias'], T(prop), 0.1)
    gr = {k: torch.zeros_like(v) for k, v in t.items()}
    pred, loss = L.mf_train_fwdbwd(ctx, m, T(X), T(Y), 0, gr['uid_embeddings.weight'], gr['iid_embeddings.weight'],
                                   gr['user_bias.weight'].view(-1), gr['item_bias.weight'].view(-1), gr['global_bias'])
    close(pred, pred_o, FWD_RTOL, FWD_ATOL, 'mse pred')
    close(loss, np.asarray(loss_o).reshape(1), FWD_RTOL, FWD_ATOL, 'mse loss')
    for k in P:
        close(gr[k].view(np.shape(go[k]) or (1,)), np.asarray(go[k]).reshape(np.shape(go[k]) or (1,)), GRAD_RTOL, GRAD_ATOL, 'mse grad ' + k)


@pytest.mark.parametrize('name', ['gd', 'adagrad', 'adam'])
def test_dense_optimizer_matches_reference(L, name):
    g = load_golden('opt_' + name)
    keys = [k[5:] for k in g if k.startswith('init/')]
    fp = FlatParams({k: g['init/' + k] for k in keys})
    for s in range(int(g['steps'])):
        pre = 's%d/' % s
        for k in keys:
            fp.gviews[k].copy_(T(g[pre + 'sparse/' + k]))
        L.dense_opt_step(name, fp.p, fp.g, fp.s1, fp.s2, float(g['lr']), float(g['l2']), float(g['l2']), 50.0, s + 1,
                         zero_grad=(s % 2 == 0))
        if s % 2 == 1:
            fp.g.zero_()
        for k in keys:
            close(fp.views[k], g[pre + 'after/' + k], 1e-5, 1e-7, name + ' after ' + k)


def test_adam_element_function_equals_ieee(L):
    """The element function every optimizer kernel uses (Adam's divisions as bare Newton steps on v_rcp_f32 / a host-side
    reciprocal, the bare v_sqrt_f32: opt_device.hpp) against the same step on the compiler's IEEE division and square root:
    bit-equal (a) for Adam's denominator over EVERY non-negative finite second moment, at step numbers across the range of
    sqrt(1 - 0.999^t), and (b) for whole element steps on states spread over many orders of magnitude."""
    d = dev()
    bits = lambda t: t.view(torch.int32)
    # (a) exhaustive in v: all 2^31 - 2^23 non-negative finite floats, 2^26 at a time
    chunk = 1 << 26
    for step in (1, 2, 3, 7, 40, 333, 1000, 2500, 6000, 20000, 10 ** 6):
        for c0 in range(0, 0x7f800000, chunk):
            v = torch.arange(c0, min(c0 + chunk, 0x7f800000), dtype=torch.int32, device=d).view(torch.float32)
            n = v.numel()
            z = torch.zeros(n, device=d)
            ga, gb = torch.empty(n, device=d), torch.empty(n, device=d)
            L.debug_opt_elem('adam', 2, z, ga, z, v, 1e-3, 1e-4, 1e-4, 50.0, step)
            L.debug_opt_elem('adam', 3, z, gb, z, v, 1e-3, 1e-4, 1e-4, 50.0, step)
            assert torch.equal(bits(ga), bits(gb)), (step, c0)
            if step > 3:
                break                      # (every v at the first steps — where c is far from 1 — and the low 2^26 at the others)
    # (b) whole steps: parameters, gradients and moments log-uniform over wide ranges, exact zeros mixed in
    gen = torch.Generator(device=d).manual_seed(5)
    n = 1 << 24

    def spread(lo, hi, zero_frac=0.02, signed=True):
        e = torch.rand(n, generator=gen, device=d) * (hi - lo) + lo
        x = torch.pow(10.0, e) * (1.0 + torch.rand(n, generator=gen, device=d))
        if signed:
            x = x * (torch.randint(0, 2, (n,), generator=gen, device=d).float() * 2 - 1)
        return torch.where(torch.rand(n, generator=gen, device=d) < zero_frac, torch.zeros_like(x), x).contiguous()

    for step, lr, l2, grad in ((1, 1e-3, 1e-4, True), (2, 1e-3, 1e-4, False), (9, 0.1, 0.0, True), (1500, 1e-3, 1e-4, False),
                               (100000, 1e-2, 1e-2, True), (7, 1e-3, 1e-4, False)):
        p0 = spread(-12, 1)
        g0 = spread(-10, 3) if grad else torch.zeros(n, device=d)           # (beyond +-50: the clip)
        m0 = spread(-14, 2)
        v0 = (m0 * m0 * spread(-3, 3, 0.0, signed=False)).contiguous()
        v0 = torch.where(torch.rand(n, generator=gen, device=d) < 0.02, torch.zeros_like(v0), v0)
        outs = []
        for ieee in (0, 1):
            st = [p0.clone(), g0.clone(), m0.clone(), v0.clone()]
            for k in range(3):                                               # three steps in a row: the states feed back
                L.debug_opt_elem('adam', ieee, st[0], st[1], st[2], st[3], lr, l2, l2, 50.0, step + k)
            outs.append(st)
        for a, b, nm in zip(outs[0], outs[1], 'pgmv'):
            assert torch.isfinite(a).all()
            assert torch.equal(bits(a), bits(b)), (step, nm, int((bits(a) != bits(b)).sum()))
        # the four-at-a-time form of the float4 kernels (its l2 head is written on 2-vectors): same bits
        st = [p0[:n - 3].clone(), g0[:n - 3].clone(), m0[:n - 3].clone(), v0[:n - 3].clone()]       # (n - 3: a scalar tail too)
        for k in range(3):
            L.debug_opt_elem('adam', 4, st[0], st[1], st[2], st[3], lr, l2, l2, 50.0, step + k)
        for a, b, nm in zip(outs[0], st, 'pgmv'):
            assert torch.equal(bits(a[:n - 3]), bits(b)), (step, nm, 'opt_elem4')
    # the other optimizers share nothing with the change, but go through the same entry
    for kind in ('gd', 'adagrad'):
        outs = []
        for ieee in (0, 1):
            st = [p0.clone(), g0.clone(), m0.abs().clone(), v0.clone()]
            L.debug_opt_elem(kind, ieee, st[0], st[1], st[2], st[3], 1e-2, 1e-4, 1e-4, 50.0, 3)
            outs.append(st)
        assert all(torch.equal(bits(a), bits(b)) for a, b in zip(outs[0], outs[1]))


def test_dense_optimizer_tail_and_large(L):
    """n not a multiple of 4 and n larger than one grid sweep, against the oracle's optimizer."""
    rng = np.random.RandomState(2)
    for n in (1, 7, 1027, 3 * 1024 * 1024 + 5):
        p0 = rng.randn(n).astype(np.float32)
        g0 = (rng.randn(n) * 30).astype(np.float32)
        n_pad = (n + 3) // 4 * 4
        buf = torch.zeros(4, n_pad, dtype=torch.float32, device=dev())
        buf[0, :n] = T(p0)
        opt = O.DenseOptimizer('adam', 0.01, 1e-3)
        P = {'p': p0.copy()}
        for step in (1, 2):
            buf[1, :n] = T(g0)
            L.dense_opt_step('adam', buf[0, :n], buf[1, :n], buf[2, :n], buf[3, :n], 0.01, 1e-3, 1e-3, 50.0, step)
            P, _ = O.train_step(P, opt, 1e-3, {'p': g0})
            close(buf[0, :n], P['p'], 1e-5, 1e-7, 'adam n=%d step %d' % (n, step))


def test_train_negative_sampler_matches_oracle(L):
    rng = np.random.RandomState(4)
    U_, I_ = 300, 120
    uids, hist = [], []
    indptr = [0]
    for u in range(U_):
        deg = int(rng.randint(0, 40)) if u != 7 else 110      # user 7: fewer than 20 % of the items remain
        items = np.sort(rng.choice(I_, size=min(deg, I_ - 2), replace=False))
        hist.append(items)
        indptr.append(indptr[-1] + len(items))
        k = len(items) if u != 7 else 5
        uids.extend([u] * k)
    uids = np.array(uids, dtype=np.int64)
    perm = rng.permutation(len(uids))
    uids = uids[perm]                                  # sample-id order
    hist_items = np.concatenate(hist).astype(np.int64)
    hist_indptr = np.array(indptr, dtype=np.int64)
    order = np.argsort(uids, kind='stable')
    rows = order.astype(np.int64)                      # rows of each user in ascending sample id
    rows_indptr = np.searchsorted(uids[order], np.arange(U_ + 1)).astype(np.int64)
    for epoch in (0, 3):
        out = L.sample_train_negatives(T(rows_indptr), T(rows), T(hist_indptr), T(hist_items), U_, I_, 2019, epoch).cpu().numpy()
        ref = PH.train_negatives(2019, epoch, uids, I_, hist_indptr, hist_items)
        assert np.array_equal(out, ref)
        for u in range(U_):
            mine = out[uids == u]
            assert len(set(mine.tolist())) == len(mine)                      # distinct within the epoch
            assert not set(mine.tolist()) & set(hist[u].tolist())            # never a train positive
        assert 0 not in out[uids == 7]                                       # low-remaining regime skips item 0
    a = L.sample_train_negatives(T(rows_indptr), T(rows), T(hist_indptr), T(hist_items), U_, I_, 2019, 0).cpu().numpy()
    b = L.sample_train_negatives(T(rows_indptr), T(rows), T(hist_indptr), T(hist_items), U_, I_, 2019, 1).cpu().numpy()
    assert not np.array_equal(a, b)


def test_train_negative_sampler_exhausted_users_terminate(L):
    """in the low regime (< 20 % of the items remain) item 0 is never drawn (DataProcessor.py:490-493); a user whose
    only remaining item is 0 must get -1 (the reference raises there) instead of spinning, and a user with exactly one
    admissible item must get it."""
    I_ = 120
    hists = [np.arange(1, I_), np.arange(0, I_ - 1), np.arange(0, I_), np.array([3, 5])]      # only 0 left; only 119 left; nothing; plenty
    nrows = [1, 2, 1, 3]
    hist_indptr = np.concatenate([[0], np.cumsum([len(h) for h in hists])]).astype(np.int64)
    hist_items = np.concatenate(hists).astype(np.int64)
    uids = np.repeat(np.arange(4), nrows).astype(np.int64)
    rows = np.arange(len(uids), dtype=np.int64)
    rows_indptr = np.searchsorted(uids, np.arange(5)).astype(np.int64)
    out = L.sample_train_negatives(T(rows_indptr), T(rows), T(hist_indptr), T(hist_items), 4, I_, 2019, 1).cpu().numpy()
    ref = PH.train_negatives(2019, 1, uids, I_, hist_indptr, hist_items)
    assert np.array_equal(out, ref)
    assert out[0] == -1 and out[1] == I_ - 1 and out[2] == -1 and out[3] == -1
    assert set(out[4:].tolist()).isdisjoint({3, 5}) and len(set(out[4:].tolist())) == 3 and out[4:].min() >= 0


def test_argument_errors_are_reported(L, ctx):
    t = torch.zeros(4, 16, device=dev())
    with pytest.raises(RuntimeError):
        L.dense_opt_step('adam', t.view(-1), t.view(-1), None, None, 0.1, 0, 0, 50, 1)
    with pytest.raises(RuntimeError):
        L.dense_opt_step('adam', torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4), 0.1, 0, 0, 50, 1)  # CPU tensors
    W = torch.zeros(260, 260 + 8, device=dev())      # wider than the largest column tile (256 since round 3)
    m = L.model_struct(torch.zeros(5, 260, device=dev()), torch.zeros(5, 260, device=dev()), W, torch.zeros(260, device=dev()),
                       torch.zeros(5, 8, device=dev()), torch.zeros(5, 5, device=dev()), 10, 2, 0.1)
    with pytest.raises(RuntimeError, match='D must be'):
        L.dccf_predict(ctx, m, L.rand_struct(seed=1), torch.zeros(2, 2, dtype=torch.int64, device=dev()), 0.0)
    # gradients of an extra mlp layer missing
    Wx, bx = torch.zeros(16, 16, device=dev()), torch.zeros(16, device=dev())
    m2 = L.model_struct(torch.zeros(5, 16, device=dev()), torch.zeros(5, 16, device=dev()), torch.zeros(16, 24, device=dev()),
                        torch.zeros(16, device=dev()), torch.zeros(5, 8, device=dev()), torch.zeros(5, 5, device=dev()), 10, 2, 0.1,
                        extra=[(Wx, bx)])
    with pytest.raises(RuntimeError, match='extra mlp layer'):
        L.dccf_train_fwdbwd(ctx, m2, L.rand_struct(seed=1), torch.zeros(2, 2, dtype=torch.int64, device=dev()),
                            torch.zeros(2, device=dev()), 1, 0.0, torch.zeros(5, 16, device=dev()), torch.zeros(5, 16, device=dev()),
                            torch.zeros(16, 24, device=dev()), torch.zeros(16, device=dev()))
    # empty batch is legal
    m = L.model_struct(torch.zeros(5, 16, device=dev()), torch.zeros(5, 16, device=dev()), torch.zeros(16, 24, device=dev()),
                       torch.zeros(16, device=dev()), torch.zeros(5, 8, device=dev()), torch.zeros(5, 5, device=dev()), 10, 2, 0.1)
    out = L.dccf_predict(ctx, m, L.rand_struct(seed=1), torch.zeros(0, 2, dtype=torch.int64, device=dev()), 0.0)
    assert out.shape == (0,)


def test_large_batch_properties(L, ctx):
    """At eval size (N = 16384 rows, the reference's eval_batch_size) the oracle is too slow; check size-independent
    properties: determinism of the fused draws, a different step gives different draws, and splitting the batch
    gives the same predictions row by row (rows are independent given the counters' row index)."""
    rng = np.random.RandomState(6)
    U_, I_, D, F, S, A, N = 2000, 3000, 64, 768, 10, 2, 16384
    Pm = [T((rng.randn(U_, D) * 0.3).astype(np.float32)), T((rng.randn(I_, D) * 0.3).astype(np.float32)),
          T((rng.randn(D, D + F) * 0.05).astype(np.float32)), T((rng.randn(D) * 0.1).astype(np.float32))]
    feat = T((rng.randn(I_, F) * 0.5).astype(np.float32))
    expo = T(rng.randn(U_, I_).astype(np.float32))
    m = L.model_struct(Pm[0], Pm[1], Pm[2], Pm[3], feat, expo, S, A, 0.1)
    X = T(np.stack([rng.randint(0, U_, N), rng.randint(0, I_, N)], 1).astype(np.int64))
    a = L.dccf_predict(ctx, m, L.rand_struct(seed=5, step=9), X, 0.0).clone()
    b = L.dccf_predict(ctx, m, L.rand_struct(seed=5, step=9), X, 0.0).clone()
    c = L.dccf_predict(ctx, m, L.rand_struct(seed=5, step=10), X, 0.0).clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert bool(torch.isfinite(a).all())
    head = L.dccf_predict(ctx, m, L.rand_struct(seed=5, step=9), X[:4096].contiguous(), 0.0)
    assert torch.equal(head, a[:4096])
    # Monte-Carlo consistency: the mean over many noise draws approaches the noise-free score's neighbourhood
    close(a.mean().reshape(1), c.mean().reshape(1).cpu().numpy(), 5e-2, 1e-3, 'step-to-step mean')


def test_tables_beyond_2g_elements_equal_compact_tables(L):
    """BASELINE config 5 territory: a user table of 17 M x 128 = 2.2e9 elements (flat offsets beyond 2^31, 8.7 GB) cannot be
    checked by the oracle, so it is checked by a size-independent property: two training steps (forward, BPR, backward,
    dense l2 + GD step, second step prepared by the first) on the big table must equal the same steps on a COMPACT table
    that holds only the users the batches touch — the fused draws depend on (seed, step, batch row), not on the table.
    Touched rows, V, W, b and the predictions agree to the float-atomic tolerance; an untouched row at the far end of the
    big table moved by exactly the l2 term."""
    from dccf_amd.models import DCCF, FusedOptimizer
    if torch.cuda.get_device_properties(0).total_memory < 100e9:
        pytest.skip('needs ~30 GB of HBM')
    U, I, D, F, S, A, B = 17_000_000, 5000, 128, 64, 10, 2, 64
    g = torch.Generator(device='cuda').manual_seed(3)
    feat = torch.randn(I, F, generator=g, device='cuda') * 0.3
    ips = dict(P=torch.randn(U, 4, generator=g, device='cuda') * 0.3, Q=torch.randn(I, 4, generator=g, device='cuda') * 0.3,
               bu=torch.randn(U, generator=g, device='cuda') * 0.1, bi=torch.randn(I, generator=g, device='cuda') * 0.1,
               prop=torch.rand(I, generator=g, device='cuda'), b0=0.1, M=0.1)
    # users from the whole range, the last rows of the table included (element offsets up to 2.176e9)
    us = torch.cat([torch.randint(0, U, (2 * B - 3,), generator=g, device='cuda'),
                    torch.tensor([U - 1, U - 2, 16_900_000], device='cuda')])
    full = torch.stack([torch.stack([us[torch.randperm(2 * B, generator=g, device='cuda')],
                                     torch.randint(0, I, (2 * B,), generator=g, device='cuda')], 1) for _ in range(2)])
    uniq, inv = torch.unique(full[:, :, 0], return_inverse=True)
    lr, l2 = 0.05, 0.01
    res = []
    for mode in ('big', 'compact'):
        Um = U if mode == 'big' else uniq.numel()
        sub = (lambda t: t) if mode == 'big' else (lambda t: t[uniq].contiguous())
        m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=S, attribute_num=A, std=0.1, label_min=0, label_max=1,
                 feature_num=0, user_num=Um, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=1, random_seed=31,
                 model_path='/tmp/x.pt', feature_embedding=feat,
                 ips_factors=dict(ips, P=sub(ips['P']), bu=sub(ips['bu'])))
        torch.manual_seed(1)
        m.apply(m.init_paras)
        P = dict(m.named_parameters())
        if mode == 'big':
            big_rows = P['uid_embeddings.weight'].detach()[uniq].clone()
            keep = [P[k].detach().clone() for k in ('iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias')]
            far = P['uid_embeddings.weight'].detach()[U - 3].clone()          # never in a batch
        else:
            with torch.no_grad():
                P['uid_embeddings.weight'].copy_(big_rows)
                for k, v in zip(('iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias'), keep):
                    P[k].copy_(v)
        m.optimizer = FusedOptimizer(m, 'gd', lr, l2)
        m.train()
        y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')])
        Xs = full if mode == 'big' else torch.stack([inv, full[:, :, 1]], 2).contiguous()
        preds = []
        for k in range(2):
            out = m.train_step({'X': Xs[k], 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.2}, X_next=Xs[k + 1] if k == 0 else None)
            preds.append(out['prediction'].clone())
        m.optimizer.flush()       # the views in P were taken before training: bring the rows the lazy regularisation left behind
        torch.cuda.synchronize()
        assert m.ctx.prepared_steps() == 1 and m.optimizer.lazy is not None and not m.optimizer.lazy.dirty
        Uw = P['uid_embeddings.weight'].detach()
        res.append((Uw[uniq].clone() if mode == 'big' else Uw.clone(), P['iid_embeddings.weight'].detach().clone(),
                    P['mlp.0.weight'].detach().clone(), P['mlp.0.bias'].detach().clone(), preds))
        if mode == 'big':
            want = far.clone()
            for _ in range(2):     # two GD steps of the l2 term alone: p -= lr * (2 l2 p + l2 p)
                want = want - lr * ((2.0 * l2) * want + l2 * want)
            np.testing.assert_allclose(Uw[U - 3].cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-9)
            assert int(m.touchedU.sum()) == 0 and float(m.flat_g[-4096:].abs().max()) == 0.0
        del m, P, Uw
        torch.cuda.empty_cache()
    for a, b in zip(res[0][:4], res[1][:4]):
        close(b, a.cpu().numpy(), 0, 2e-6, 'big table vs compact table')
    for a, b in zip(res[0][4], res[1][4]):
        close(b, a.cpu().numpy(), 1e-5, 1e-6, 'predictions')


@pytest.mark.parametrize('direct', [True, False])
def test_sharded_trainer_with_hip_backend_world1(L, direct):
    """dccf_amd/sharded.py with the real HIP backend over RCCL at world size 1 (the routing at world size 2 is covered
    on CPU by tests/test_sharded_gloo.py): two steps must equal the oracle on the same counter-based draws — with the
    collectives straight on RCCL (dccf_comm_*, the default under an RCCL process group) and through torch.distributed."""
    import os
    import torch.distributed as dist
    from dccf_amd.sharded import ShardedDCCF, HipBackend
    from test_sharded_gloo import make_world, expo_from_ips, CFG, KEYS
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    from conftest import free_port
    os.environ['MASTER_PORT'] = str(free_port())
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev())
    try:
        c = dict(CFG)
        P, feat, ips, X = make_world(c)
        ips_t = dict(P=T(ips['P']), bu=T(ips['bu']), Q=T(ips['Q']), bi=T(ips['bi']), prop=T(ips['prop']), b0=0.1, M=0.1)
        tr = ShardedDCCF(0, 1, c['U'], c['I'], c['D'], c['S'], c['A'], c['std'], c['dropout'], c['lr'], c['l2'], c['seed'],
                         HipBackend(dev()), dev(), T(feat), ips_t, direct=direct)
        assert (tr.comm is not None) == direct
        tr.set_global_params(T(P[KEYS[0]]), T(P[KEYS[1]]), T(P[KEYS[2]]), T(P[KEYS[3]]))
        expo = expo_from_ips(ips)
        opt = O.DenseOptimizer('adam', c['lr'], c['l2'])
        N, Ld = 2 * c['B'], 2 * c['B'] * (c['S'] + 1) * c['A']
        Y = np.concatenate([np.ones(c['B'], np.float32), np.zeros(c['B'], np.float32)])
        tr.begin_epoch(T(np.stack([x[:1] for x in X])), 3)     # rank 0's batches only (world size 1), epoch 3
        cand_all = PH.candidates(c['seed'], 3, c['steps'] * N, c['S'], c['I']).reshape(c['steps'], N, c['S'])
        for step in range(c['steps']):
            x = X[step][:1]
            pred, loss = tr.train_step(step)
            cand = cand_all[step]
            noise = L.debug_noise(Ld, c['F'], c['std'], c['seed'], step, dev()).cpu().numpy()
            keep = PH.dropout_keep(c['seed'], step, Ld, c['D'], float(np.float32(c['dropout'])))
            fw = O.dccf_forward(P, feat, expo, x[0], cand, noise, keep, c['dropout'], c['A'])
            close(pred, fw['prediction'], FWD_RTOL, FWD_ATOL, 'sharded pred')
            _, dpred = O.loss_and_dpred(fw['prediction'], Y, 1)
            P, _ = O.train_step(P, opt, c['l2'], O.dccf_backward(P, fw, dpred, c['A']))
        tr.flush()             # (the shard's lazy regularisation: rows no step touched are brought up to date)
        for k, t in zip(KEYS, (tr.U, tr.V, tr.W, tr.b)):
            close(t, P[k], PARAM_RTOL, PARAM_ATOL + STEP_FRAC * c['lr'], 'sharded param ' + k)
    finally:
        dist.destroy_process_group()


def test_row_aware_optimizer_equals_dense(L, ctx):
    """dccf_dense_opt_step_rows (skips the gradient of rows whose "touched" byte is 0) must give bit-identical
    parameters to the dense step, over several training steps with the backward setting the bytes."""
    rng = np.random.RandomState(12)
    U_, I_, D, F, S, A, pairs = 700, 520, 64, 160, 10, 2, 24
    shapes = [(U_, D), (I_, D), (D, D + F), (D,)]
    sizes = [int(np.prod(s)) for s in shapes]
    pads = [(n + 255) // 256 * 256 for n in sizes]
    init = (rng.randn(sum(pads)) * 0.1).astype(np.float32)
    feat, expo = T((rng.randn(I_, F) * 0.5).astype(np.float32)), T(rng.randn(U_, I_).astype(np.float32))
    runs = []
    for rows in (False, True):
        buf = [T(init.copy()), torch.zeros(sum(pads), device=dev()), torch.zeros(sum(pads), device=dev()),
               torch.zeros(sum(pads), device=dev())]
        offs = np.cumsum([0] + pads[:-1])
        v = [buf[0][o:o + n].view(s) for o, n, s in zip(offs, sizes, shapes)]
        gv = [buf[1][o:o + n].view(s) for o, n, s in zip(offs, sizes, shapes)]
        tU = torch.zeros(U_, dtype=torch.uint8, device=dev())
        tV = torch.zeros(I_, dtype=torch.uint8, device=dev())
        m = L.model_struct(v[0], v[1], v[2], v[3], feat, expo, S, A, 0.1)
        r2 = np.random.RandomState(99)
        for step in range(1, 5):
            u = r2.randint(0, U_, pairs)
            X = T(np.concatenate([np.stack([u, r2.randint(0, I_, pairs)], 1), np.stack([u, r2.randint(0, I_, pairs)], 1)]).astype(np.int64))
            Y = torch.cat([torch.ones(pairs, device=dev()), torch.zeros(pairs, device=dev())])
            L.dccf_train_fwdbwd(ctx, m, L.rand_struct(seed=5, step=step), X, Y, 1, 0.2, gv[0], gv[1], gv[2], gv[3],
                                touchedU=tU if rows else None, touchedV=tV if rows else None)
            if rows:
                nU, nV = int(tU.sum()), int(tV.sum())
                assert 0 < nU <= pairs and pairs <= nV <= 2 * pairs * (S + 1)
                assert bool(((gv[0].abs().sum(1) > 0) <= (tU > 0)).all())       # a non-zero gradient row is always flagged
                L.dense_opt_step_rows('adam', buf[0], buf[1], buf[2], buf[3], 1e-3, 1e-4, 1e-4, 50.0, step,
                                      [(int(offs[0]), U_, D, tU), (int(offs[1]), I_, D, tV)])
                assert int(tU.sum()) == 0 and int(tV.sum()) == 0 and float(buf[1].abs().max()) == 0.0
            else:
                L.dense_opt_step('adam', buf[0], buf[1], buf[2], buf[3], 1e-3, 1e-4, 1e-4, 50.0, step)
        runs.append(buf[0].cpu().numpy())
    # float atomics may reorder the sums inside the backward between the two runs (Adam turns a last-bit change of a tiny
    # gradient into a fraction of lr); the optimizer arithmetic itself is the same code
    close(runs[1], runs[0], 1e-6, 4 * STEP_FRAC * 1e-3, 'row-aware vs dense optimizer')


def test_graph_replayed_steps_equal_eager_steps(L):
    """StepGraph (device-side step counter, one hipGraph replay per step, eager partial tail) must train exactly like the
    eager step-by-step path: same batches, same Philox counters, same Adam step numbers."""
    from dccf_amd.models import DCCF, FusedOptimizer, StepGraph
    rng = np.random.RandomState(21)
    U_, I_, D, F, B, nb = 400, 300, 64, 160, 16, 5
    feat = T((rng.randn(I_, F) * 0.5).astype(np.float32))
    expo = T(rng.randn(U_, I_).astype(np.float32))
    full_np = np.zeros((2, nb, 2 * B, 2), dtype=np.int64)               # two epochs
    for e in range(2):
        for k in range(nb):
            u = rng.randint(0, U_, B)
            full_np[e, k] = np.concatenate([np.stack([u, rng.randint(0, I_, B)], 1), np.stack([u, rng.randint(0, I_, B)], 1)])
    tail_np = full_np[0, 0, [0, 1, 2, B, B + 1, B + 2]]                     # a 3-pair partial batch
    results = []
    for use_graph in (False, True):
        m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=10, attribute_num=2, std=0.1, label_min=0,
                 label_max=1, feature_num=0, user_num=U_, item_num=I_, u_vector_size=D, i_vector_size=D, n_layers=1,
                 random_seed=2019, model_path='/tmp/x.pt', feature_embedding=feat, expo_prob=expo)
        torch.manual_seed(1)
        m.apply(m.init_paras)
        opt = FusedOptimizer(m, 'adam', 1e-3, 1e-4)
        m.optimizer = opt
        m.train()
        if use_graph:
            sg = StepGraph(m, opt, nb, 2 * B, 0.2)
            for e in range(2):
                sg.load_epoch(T(full_np[e]))
                for _ in range(nb):
                    sg.step()
                sg.tail(T(tail_np))
        else:
            y = torch.cat([torch.ones(B, device=dev()), torch.zeros(B, device=dev())])
            for e in range(2):
                for k in range(nb):
                    m({'X': T(full_np[e, k]), 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.2})
                    opt.step()
                m({'X': T(tail_np), 'Y': torch.cat([y[:3], y[B:B + 3]]), 'rank': 1, 'train': True, 'dropout': 0.2})
                opt.step()
        assert m._call == 2 * (nb + 1) and opt.t == 2 * (nb + 1)
        m.optimizer.flush()       # (rows the lazy regularisation left behind)
        results.append(m.flat_p.cpu().numpy().copy())
    # float atomics reorder sums inside the backward; Adam turns that into <= a fraction of one step (see STEP_FRAC)
    close(results[1], results[0], PARAM_RTOL, PARAM_ATOL + STEP_FRAC * 1e-3, 'graph vs eager parameters')


def test_large_batch_forward_equals_small_batches(L, ctx):
    """An eval-size batch (workgroups loop over several tiles) must give, row for row, what the same rows give in small
    batches on the same injected draws — and its fused draws must equal its injected draws bit for bit."""
    rng = np.random.RandomState(31)
    U_, I_, D, F, S, A, N = 900, 700, 64, 768, 10, 2, 3072          # L = 67,584 rows -> 2,112 tiles on 1,024 workgroups
    Pm = [T((rng.randn(U_, D) * 0.3).astype(np.float32)), T((rng.randn(I_, D) * 0.3).astype(np.float32)),
          T((rng.randn(D, D + F) * 0.05).astype(np.float32)), T((rng.randn(D) * 0.1).astype(np.float32))]
    feat = T((rng.randn(I_, F) * 0.5).astype(np.float32))
    expo = T(rng.randn(U_, I_).astype(np.float32))
    m = L.model_struct(Pm[0], Pm[1], Pm[2], Pm[3], feat, expo, S, A, 0.1)
    X = T(np.stack([rng.randint(0, U_, N), rng.randint(0, I_, N)], 1).astype(np.int64))
    seed, step, p = 99, 4, 0.2
    Ld = N * (S + 1) * A
    si = L.debug_candidates(N, S, I_, seed, step, dev())
    nz = L.debug_noise(Ld, F, 0.1, seed, step, dev())
    kp = L.debug_keep(Ld, D, p, seed, step, dev())
    big_fused = L.dccf_predict(ctx, m, L.rand_struct(seed=seed, step=step), X, p).clone()
    big_inj = L.dccf_predict(ctx, m, L.rand_struct(sample_item=si, noise=nz, keep=kp), X, p).clone()
    assert torch.equal(big_fused, big_inj)
    rows = (S + 1) * A
    parts = []
    for n0 in range(0, N, 768):                                       # 768 rows -> 528 tiles: one tile per workgroup
        n1 = n0 + 768
        r = L.rand_struct(sample_item=si[n0:n1].contiguous(), noise=nz[n0 * rows:n1 * rows].contiguous(),
                          keep=kp[n0 * rows:n1 * rows].contiguous())
        parts.append(L.dccf_predict(ctx, m, r, X[n0:n1].contiguous(), p).clone())
    assert torch.equal(big_inj, torch.cat(parts))                     # same kernel, same per-row arithmetic: bit-identical


@pytest.mark.parametrize('opt_name,B,l2,D,NL', [('gd', 128, 0.05, 64, 1), ('gd', 600, 0.05, 64, 1), ('adam', 128, 1e-4, 64, 1),
                                                ('adagrad', 37, 1e-4, 64, 1), ('gd', 64, 0.05, 128, 1), ('gd', 50, 0.05, 16, 1),
                                                # --n_layers > 1 and widths that are not a kernel tile go through the same entry
                                                ('gd', 128, 0.05, 64, 2), ('gd', 40, 0.05, 48, 3), ('gd', 50, 0.05, 24, 1)])
def test_overlapped_train_step_equals_split_step(L, opt_name, B, l2, D, NL):
    """dccf_train_step with the untouched-row optimizer pass on the side stream ('overlap'), hosted as extra workgroups of
    the backward launch ('hosted'), or with every step prepared by the one before ('prep') == forward/backward followed by the
    row-aware dense step over several steps (duplicate users/items inside a batch included).  Same arithmetic per
    element; the only run-to-run difference is the order of the float atomics inside the backward.  So: rows no batch
    touched are bit-identical; under GD (no amplification; l2 large enough that one missed or doubled row update would
    show) everything agrees to 1e-7; under Adam / Adagrad a last-bit change of a ~1e-12 gradient (a candidate with a
    vanishing exposure weight) flips sign(g) and moves a whole row by a fraction of lr — in ANY two runs of the same
    code path, measured — so there the bulk must agree and at most a few rows may sit within lr * steps."""
    from dccf_amd.models import DCCF, FusedOptimizer
    U, I, F = 3001, 1999, 96               # not multiples of 4: the word-wise marking must respect the array ends
    g = torch.Generator(device='cuda').manual_seed(5)
    feat = torch.randn(I, F, generator=g, device='cuda') * 0.05
    expo = torch.randn(U, I, generator=g, device='cuda')
    states = []
    gen = torch.Generator(device='cuda').manual_seed(9)
    full = torch.stack([torch.stack([torch.randint(0, U // 2 if k % 2 else 40, (2 * B,), generator=gen, device='cuda'),
                                     torch.randint(0, I, (2 * B,), generator=gen, device='cuda')], 1) for k in range(6)])
    tile = D in (16, 32, 64, 128)       # the form hosted in the backward launch needs one of the four tile widths; the others any multiple of 4
    # 'step' / 'prep' run with the windowed lazy regularisation (DCCF.lazy_K = 16 by default: 6 steps never complete a cycle of
    # windows, so most rows are brought up to date by the flush); 'dense' / 'denseprep' are the same calls with lazy_K = 0;
    # 'lazy3' cycles the windows twice
    for mode in ('split', 'step', 'overlap', 'prep', 'hosted', 'dense', 'denseprep', 'lazy3') if tile else ('split', 'step', 'overlap', 'prep', 'dense', 'denseprep', 'lazy3'):
        m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=10, attribute_num=2, std=0.1, label_min=0, label_max=1,
                 feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=NL, random_seed=11,
                 model_path='/tmp/x.pt', feature_embedding=feat, expo_prob=expo)
        torch.manual_seed(3)
        m.apply(m.init_paras)
        m.optimizer = FusedOptimizer(m, opt_name, 0.01, l2)
        m.lazy_K = {'dense': 0, 'denseprep': 0, 'lazy3': 3}.get(mode, 16)
        m.train()
        y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')])
        preds = []
        seen = torch.zeros(U, dtype=torch.bool, device='cuda')
        for k in range(6):
            X = full[k]
            seen[X[:, 0]] = True
            batch = {'X': X, 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.2}
            if mode == 'split':
                out = m(batch)
                m.optimizer.step()
            elif mode in ('prep', 'denseprep', 'lazy3'):      # the optimizer launch of step k draws step k + 1's candidates and writes W^T
                out = m.train_step(batch, X_next=full[k + 1] if k < 5 else None)
            else:
                out = m.train_step(batch, overlap={'overlap': 1, 'hosted': 2}.get(mode, 0))
            preds.append(out['prediction'].clone())
        lazy_on = D % 4 == 0 and mode in ('step', 'prep', 'lazy3')
        assert (m.optimizer.lazy is not None) == lazy_on
        if lazy_on:       # rows are behind until the flush (model.eval() / state_dict() / l2() / any dense call do it)
            assert m.optimizer.lazy.dirty and int(m.optimizer.lazy.last.min()) < 6
            m.eval()
            assert not m.optimizer.lazy.dirty and int(m.optimizer.lazy.last.min()) == 6 and int(m.optimizer.lazy.last.max()) == 6
        torch.cuda.synchronize()
        assert m.ctx.prepared_steps() == (5 if mode in ('prep', 'denseprep', 'lazy3') else 0)
        assert m.touchedU is None or (int(m.touchedU.sum()) == 0 and int(m.touchedV.sum()) == 0)
        assert float(m.flat_g.abs().max()) == 0.0
        assert len(m.state_dict()) == 2 + 2 * NL
        states.append([m.flat_p.clone(), m.optimizer.s1, m.optimizer.s2, seen] + preds)
    for other in states[1:]:
        assert torch.equal(states[0][3], other[3])
        never = ~states[0][3]                       # user rows no batch touched: only the l2 term moved them
        for a, b in zip(states[0][:3], other[:3]):
            if a is not None:
                assert torch.equal(a[:U * D].view(U, D)[never], b[:U * D].view(U, D)[never])
                if opt_name == 'gd':
                    close(b, a.cpu().numpy(), 0, 1e-7, 'overlapped vs split step')
                else:
                    d = (a - b).abs()
                    assert int((d > 6 * STEP_FRAC * 0.01).sum()) <= 4 * D + 8 and float(d.max()) <= 6 * 0.01
        close(other[4], states[0][4].cpu().numpy(), 1e-6, 1e-7, 'first prediction')
        if opt_name == 'gd':      # same candidates, noise and W^T at every later step too
            for a, b in zip(states[0][5:], other[5:]):
                close(b, a.cpu().numpy(), 1e-4, 1e-5, 'later predictions')


@pytest.mark.parametrize('opt_name,B,D,NL,F', [('adam', 128, 64, 1, 96), ('adagrad', 37, 64, 1, 96), ('adam', 64, 128, 1, 97),
                                              ('adam', 40, 48, 3, 96), ('adagrad', 128, 64, 2, 96), ('gd', 50, 16, 1, 96)])
def test_every_form_of_the_train_step_is_bit_identical_in_deterministic_mode(L, opt_name, B, D, NL, F):
    """dccf_ctx_set_deterministic ("a sorted-segment deterministic mode for tests"): the backward stores one
    gradient row per (batch row, candidate) slot and adds the slots of a destination in slot order, dW / gb from partial sums in
    index order — no float atomic.  Then EVERY form of the step — the split calls (forward/backward, then the dense optimizer),
    dccf_train_step with the lazy regularisation (`step`), announced steps (`prep`), the side-stream and hosted overlap forms,
    the dense forms, a lazy window that cycles (`lazy3`) — must leave the same bits in every parameter, in both Adam moments and
    in every prediction: Adam / Adagrad included, where the float-atomic path can only be held to a fraction of lr."""
    from dccf_amd.models import DCCF, FusedOptimizer
    U, I = 3001, 1999
    g = torch.Generator(device='cuda').manual_seed(5)
    feat = torch.randn(I, F, generator=g, device='cuda') * 0.05
    expo = torch.randn(U, I, generator=g, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(9)
    nst = 7
    # duplicate users inside a batch (40 distinct ones on even steps) and duplicate items: the case atomics reorder
    full = torch.stack([torch.stack([torch.randint(0, U // 2 if k % 2 else 40, (2 * B,), generator=gen, device='cuda'),
                                     torch.randint(0, 60 if k % 3 == 0 else I, (2 * B,), generator=gen, device='cuda')], 1) for k in range(nst)])
    tile = D in (16, 32, 64, 128)
    states = []
    modes = ('split', 'step', 'overlap', 'prep', 'hosted', 'dense', 'denseprep', 'lazy3', 'split') if tile else ('split', 'step', 'overlap', 'prep', 'dense', 'denseprep', 'lazy3', 'split')
    for mode in modes:
        m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=10, attribute_num=2, std=0.1, label_min=0, label_max=1,
                 feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=NL, random_seed=11,
                 model_path='/tmp/x.pt', feature_embedding=feat, expo_prob=expo)
        m.ctx.set_deterministic(True)
        torch.manual_seed(3)
        m.apply(m.init_paras)
        m.optimizer = FusedOptimizer(m, opt_name, 0.01, 1e-3)
        m.lazy_K = {'dense': 0, 'denseprep': 0, 'lazy3': 3}.get(mode, 16)
        m.train()
        y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')])
        preds, losses = [], []
        for k in range(nst):
            batch = {'X': full[k], 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.2}
            if mode == 'split':
                out = m(batch)
                m.optimizer.step()
            elif mode in ('prep', 'denseprep', 'lazy3'):
                out = m.train_step(batch, X_next=full[k + 1] if k + 1 < nst else None)
            else:
                out = m.train_step(batch, overlap={'overlap': 1, 'hosted': 2}.get(mode, 0))
            preds.append(out['prediction'].clone())
            losses.append(out['loss'].clone())
        sd = m.state_dict()          # flushes
        torch.cuda.synchronize()
        states.append((mode, m.flat_p.clone(), m.optimizer.s1, m.optimizer.s2, preds, losses))
    ref = states[0]
    for other in states[1:]:
        for a, b, what in zip(ref[1:4], other[1:4], ('parameters', 'first moment / sum', 'second moment')):
            if a is not None:
                assert torch.equal(a, b), '%s: %s differ from the split calls (max %g)' % (other[0], what, float((a - b).abs().max()))
        for k, (a, b) in enumerate(zip(ref[4], other[4])):
            assert torch.equal(a, b), '%s: prediction of step %d' % (other[0], k)
        for k, (a, b) in enumerate(zip(ref[5], other[5])):
            assert torch.equal(a, b), '%s: loss of step %d' % (other[0], k)


def test_deterministic_mode_agrees_with_the_atomic_path_and_the_oracle(L, ctx):
    """The deterministic scatter is another ORDER of the same sums: against the float-atomic backward (the default) and against
    the oracle on the same fused draws the gradients agree to the float-atomic tolerance; two deterministic runs are equal."""
    rng = np.random.RandomState(3)
    U_, I_, D, F, S, A, pairs, p = 300, 200, 64, 96, 10, 2, 24, 0.2
    keys = PKEYS
    P = {keys[0]: (rng.randn(U_, D) * 0.3).astype(np.float32), keys[1]: (rng.randn(I_, D) * 0.3).astype(np.float32),
         keys[2]: (rng.randn(D, D + F) * 0.1).astype(np.float32), keys[3]: (rng.randn(D) * 0.1).astype(np.float32)}
    feat = (rng.randn(I_, F) * 0.5).astype(np.float32)
    expo = rng.randn(U_, I_).astype(np.float32)
    u = rng.randint(0, 12, pairs)                                    # heavy duplication of users and items
    X = np.concatenate([np.stack([u, rng.randint(0, 20, pairs)], 1), np.stack([u, rng.randint(0, 20, pairs)], 1)]).astype(np.int64)
    Y = np.concatenate([np.ones(pairs), np.zeros(pairs)]).astype(np.float32)
    tp = [T(P[k]) for k in keys]
    m = L.model_struct(tp[0], tp[1], tp[2], tp[3], T(feat), T(expo), S, A, 0.1)
    N, seed, step = 2 * pairs, 77, 3
    runs = []
    c2 = L.Context(0)
    for det in (False, True, True):
        c2.set_deterministic(det)
        gr = [torch.zeros_like(t) for t in tp]
        tu, tv = torch.zeros(U_ + 3, dtype=torch.uint8, device=dev())[:U_], torch.zeros(I_ + 3, dtype=torch.uint8, device=dev())[:I_]
        pred, loss = L.dccf_train_fwdbwd(c2, m, L.rand_struct(seed=seed, step=step), T(X), T(Y), 1, p, *gr, touchedU=tu, touchedV=tv)
        runs.append((gr, pred.clone(), loss.clone(), tu.clone(), tv.clone()))
    si = L.debug_candidates(N, S, I_, seed, step, dev()).cpu().numpy()
    nz = L.debug_noise(N * (S + 1) * A, F, 0.1, seed, step, dev()).cpu().numpy()
    kp = L.debug_keep(N * (S + 1) * A, D, p, seed, step, dev()).cpu().numpy()
    fw = O.dccf_forward(P, feat, expo, X, si, nz, kp, p, A)
    lo, dpred = O.loss_and_dpred(fw['prediction'], Y, 1)
    go = O.dccf_backward(P, fw, dpred, A)
    for k, a, b, c in zip(keys, runs[0][0], runs[1][0], runs[2][0]):
        assert torch.equal(b, c), k                                   # deterministic twice: the same bits
        close(b, a.cpu().numpy(), GRAD_RTOL, GRAD_ATOL, 'deterministic vs atomic ' + k)
        close(b, go[k], GRAD_RTOL, GRAD_ATOL, 'deterministic vs oracle ' + k)
    assert torch.equal(runs[1][1], runs[0][1]) and torch.equal(runs[1][3], runs[0][3]) and torch.equal(runs[1][4], runs[0][4])
    close(runs[1][2], np.asarray(lo, dtype=np.float32).reshape(1), FWD_RTOL, FWD_ATOL, 'loss')


def test_prepared_state_survives_nothing_it_should_not(L):
    """The prepared-next-step state machine under abuse: a predict between two train steps, a tail batch of another size,
    an announced batch that never comes, an announced batch replaced by another one at a different address — every
    sequence must give what the split calls (forward/backward, then the dense step) give.  GD with a large l2 so that one
    missed, doubled or misplaced row update would show; float atomics are the only tolerated difference."""
    from dccf_amd.models import DCCF, FusedOptimizer
    U, I, D, F, B = 901, 777, 64, 96, 64
    g = torch.Generator(device='cuda').manual_seed(15)
    feat = torch.randn(I, F, generator=g, device='cuda') * 0.05
    expo = torch.randn(U, I, generator=g, device='cuda')

    def batch(n):
        return torch.stack([torch.randint(0, U, (2 * n,), generator=g, device='cuda'),
                            torch.randint(0, I, (2 * n,), generator=g, device='cuda')], 1)
    Xs = [batch(B) for _ in range(6)] + [batch(23)]
    Xe = batch(200)
    # (batch index, announced next batch index or None, predict before it?)
    script = [(0, 1, False), (1, 2, True), (2, 3, False), (6, None, False), (3, 4, False), (5, 1, False), (1, None, True), (4, 0, False),
              (0, None, False)]
    states = []
    for mode in ('split', 'step'):
        m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=10, attribute_num=2, std=0.1, label_min=0, label_max=1,
                 feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=1, random_seed=11,
                 model_path='/tmp/x.pt', feature_embedding=feat, expo_prob=expo)
        torch.manual_seed(3)
        m.apply(m.init_paras)
        m.optimizer = FusedOptimizer(m, 'gd', 0.01, 0.05)
        preds = []
        for k, nxt, pred_first in script:
            if pred_first:                 # an evaluation call on the same context between two training steps
                m.eval()
                preds.append(m.predict({'X': Xe, 'rank': 1, 'train': False, 'dropout': 0.0})['prediction'].clone())
            m.train()
            X = Xs[k]
            n = X.shape[0] // 2
            y = torch.cat([torch.ones(n, device='cuda'), torch.zeros(n, device='cuda')])
            fd = {'X': X, 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.2}
            if mode == 'split':
                out = m(fd)
                m.optimizer.step()
            else:
                out = m.train_step(fd, X_next=Xs[nxt] if nxt is not None else None)
            preds.append(out['prediction'].clone())
        torch.cuda.synchronize()
        assert int(m.touchedU.sum()) == 0 and int(m.touchedV.sum()) == 0 and float(m.flat_g.abs().max()) == 0.0
        if mode == 'step':
            # 0->1 broken by the predict; 1->2 holds; 2->3 followed by the tail batch instead; 3->4 followed by batch 5
            # instead; 5->1 broken by the predict; 4->0 holds
            assert m.ctx.prepared_steps() == 2
        m.optimizer.flush()       # (rows the lazy regularisation left behind)
        states.append((m.flat_p.clone(), preds))
    a, b = states
    close(b[0], a[0].cpu().numpy(), 0, 2e-7, 'parameters after the scripted sequence')
    for pa, pb in zip(a[1], b[1]):
        close(pb, pa.cpu().numpy(), 1e-4, 1e-5, 'predictions along the sequence')


def test_prepared_next_step_state_is_what_k_prep_writes(L, ctx):
    """dccf_train_step(X_next): after the call the workspace holds exactly what k_prep would write at the start of the
    next step — cand = [item ; Philox draws of step_next], the gathered exposures, and the transposed copy of
    the UPDATED W (padding intact) — bit for bit; the next call then skips k_prep and a call with another batch does not."""
    from dccf_amd.models import DCCF, FusedOptimizer
    U, I, D, F, S, A, B = 500, 700, 64, 200, 10, 2, 96
    g = torch.Generator(device='cuda').manual_seed(2)
    feat = torch.randn(I, F, generator=g, device='cuda') * 0.05
    expo = torch.randn(U, I, generator=g, device='cuda')
    m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=S, attribute_num=A, std=0.1, label_min=0, label_max=1,
             feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=1, random_seed=21,
             model_path='/tmp/x.pt', feature_embedding=feat, expo_prob=expo)
    torch.manual_seed(1)
    m.apply(m.init_paras)
    m.optimizer = FusedOptimizer(m, 'adam', 0.01, 1e-4)
    m.train()
    full = torch.stack([torch.stack([torch.randint(0, U, (2 * B,), generator=g, device='cuda'),
                                     torch.randint(0, I, (2 * B,), generator=g, device='cuda')], 1) for _ in range(3)])
    y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')])
    batch = {'X': full[0], 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.2}
    m.train_step(batch, X_next=full[1])
    step_next = m._call + 1
    N = 2 * B
    cand, _, _ = L.debug_workspace(m.ctx, N, D, F, S, A, 0, 'cuda')
    WT, DP, FP = L.debug_workspace(m.ctx, N, D, F, S, A, 1, 'cuda')
    eg, _, _ = L.debug_workspace(m.ctx, N, D, F, S, A, 5, 'cuda')
    want = torch.cat([full[1][:, 1:2], L.debug_candidates(N, S, I, 21, step_next, 'cuda')], 1)
    assert torch.equal(cand.view(N, S + 1).long(), want)
    assert torch.equal(eg.view(N, S + 1), expo[full[1][:, 0:1].expand(N, S + 1), want])
    W = dict(m.named_parameters())['mlp.0.weight'].detach()
    ref = torch.zeros(D + FP, DP, device='cuda')
    ref[:D + F, :D] = W.t()
    assert torch.equal(WT.view(D + FP, DP), ref)
    batch['X'] = full[1]
    m.train_step(batch, X_next=full[2])
    assert m.ctx.prepared_steps() == 1
    batch['X'] = full[0]                      # not the batch that was announced: k_prep runs
    m.train_step(batch)
    assert m.ctx.prepared_steps() == 1


@pytest.mark.parametrize('D', [64, 24, 7])
def test_mf_row_aware_optimizer_equals_dense(L, D):
    """BiasedMF training steps: the dense optimizer called on the row segments (touched bytes set by k_mf_train; D = 64 takes the
    row-aware streaming pass, D = 24 — rows of whole float4 slots, but not a power of two — the plain dense pass that clears the
    bytes wholesale, D = 7 has no row segments at all) against the plain dense step without segments."""
    from dccf_amd.models import BiasedMF, FusedOptimizer
    U, I, B = 1203, 877, 96
    runs = []
    for rows in (True, False):
        m = BiasedMF(label_min=0, label_max=1, feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D,
                     random_seed=4, model_path='/tmp/mf.pt')
        torch.manual_seed(1)
        m.apply(m.init_paras)
        assert (getattr(m, 'row_segments', None) is not None) == (D % 4 == 0)
        if not rows:
            m.row_segments, m.touchedP, m.touchedQ = None, None, None
        m.optimizer = FusedOptimizer(m, 'adam', 0.01, 1e-4)
        m.train()
        gen = torch.Generator(device='cuda').manual_seed(2)
        y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')])
        for k in range(5):
            u = torch.randint(0, U, (B,), generator=gen, device='cuda')
            X = torch.stack([torch.cat([u, u]), torch.randint(0, I, (2 * B,), generator=gen, device='cuda')], 1)
            m({'X': X, 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.0})
            m.optimizer.step()
        torch.cuda.synchronize()
        assert float(m.flat_g.abs().max()) == 0.0
        if m.touchedP is not None:
            assert int(m.touchedP.sum()) == 0 and int(m.touchedQ.sum()) == 0
        m.optimizer.flush()
        runs.append(m.flat_p.clone())
    d = (runs[0] - runs[1]).abs()
    assert float(d.max()) <= 5 * 0.01 and int((d > 5 * STEP_FRAC * 0.01).sum()) <= 4 * D + 8


def test_eval_negative_sampler_matches_oracle(L):
    """dccf_sample_eval_negatives vs oracle/philox.py::eval_negatives (bit-exact) + the properties the reference's
    sampler guarantees: neg_n distinct items per user, none from the user's train / validation / test history."""
    rng = np.random.RandomState(4)
    U_, I_, neg_n = 300, 257, 30
    hist = []
    for u in range(U_):
        k = rng.randint(0, 40)
        if u == 5:
            k = 215                       # < 20 % of the items remain: item 0 is then never drawn
        if u == 7:
            k = 240                       # fewer than neg_n items remain -> -1 (the reference asserts)
        hist.append(np.sort(rng.choice(I_, k, replace=False)))
    indptr = np.r_[0, np.cumsum([len(h) for h in hist])].astype(np.int64)
    items = np.concatenate(hist).astype(np.int64)
    users = rng.permutation(U_)[:120].astype(np.int64)
    users[:2] = [5, 7]
    for tag in (1, 2):
        got = L.sample_eval_negatives(T(users), T(indptr), T(items), I_, neg_n, 99, tag).cpu().numpy()
        want = PH.eval_negatives(99, tag, users, I_, indptr, items, neg_n)
        assert np.array_equal(got, want)
        for w, u in enumerate(users):
            if u == 7:
                assert (got[w] == -1).all()
                continue
            assert len(set(got[w].tolist())) == neg_n and got[w].min() >= 0 and got[w].max() < I_
            assert not set(got[w].tolist()) & set(hist[u].tolist())
            if u == 5:
                assert 0 not in got[w]
    a = L.sample_eval_negatives(T(users), T(indptr), T(items), I_, neg_n, 99, 1).cpu().numpy()
    b = L.sample_eval_negatives(T(users), T(indptr), T(items), I_, neg_n, 99, 2).cpu().numpy()
    assert not np.array_equal(a, b)          # the two splits draw from different streams
    big = L.sample_eval_negatives(T(np.arange(50, dtype=np.int64)), T(indptr), T(items), 63001, 1000, 3, 1).cpu().numpy()
    assert big.shape == (50, 1000) and all(len(set(r.tolist())) == 1000 for r in big)


@pytest.mark.parametrize('D,F', [(64, 768), (16, 96), (128, 160)])
def test_projected_eval_tables_and_distribution(L, ctx, D, F):
    """dccf_eval_prepare / dccf_predict_projected: Pf == feat W_f^T, Lt^T Lt... (L L^T == std^2 W_f W_f^T), and the
    predictions of the projected-noise path have the distribution of the op-for-op path: for a batch that repeats the
    SAME (user, item) row, both paths produce iid samples of one random variable — their means and standard deviations
    agree within 5 standard errors, and their quartiles agree."""
    rng = np.random.RandomState(D + F)
    U_, I_, S, A, std = 40, 60, 10, 2, 0.3
    Ut, Vt = T((rng.randn(U_, D) * 0.3).astype(np.float32)), T((rng.randn(I_, D) * 0.3).astype(np.float32))
    W, b = T((rng.randn(D, D + F) * 0.1).astype(np.float32)), T((rng.randn(D) * 0.1).astype(np.float32))
    feat, expo = T((rng.randn(I_, F) * 0.5).astype(np.float32)), T(rng.randn(U_, I_).astype(np.float32))
    m = L.model_struct(Ut, Vt, W, b, feat, expo, S, A, std)
    Pf, Lt = L.dccf_eval_prepare(ctx, m)
    Wf = W[:, D:].double()
    close(Pf, (feat.double() @ Wf.T).cpu().numpy(), 2e-5, 1e-6, 'Pf')
    G = (std * std) * (Wf @ Wf.T)
    LtL = Lt.double().T @ Lt.double()                      # (L^T)^T (L^T) = L L^T
    close(LtL, G.cpu().numpy(), 2e-5, 1e-7, 'L L^T')
    assert float(torch.tril(Lt, -1).abs().max()) == 0.0     # Lt is upper triangular (L lower)
    N = 6144
    X = torch.tensor([[7, 11]], dtype=torch.int64, device=dev()).repeat(N, 1)
    full = L.dccf_predict(ctx, m, L.rand_struct(seed=3, step=1), X, 0.0).double().cpu().numpy()
    proj = L.dccf_predict_projected(ctx, m, L.rand_struct(seed=3, step=1), X, 0.0, Pf, Lt).double().cpu().numpy()
    se = np.sqrt(full.var() / N + proj.var() / N)
    assert abs(full.mean() - proj.mean()) <= 5 * se, (full.mean(), proj.mean(), se)
    assert abs(full.std() - proj.std()) <= 5 * full.std() / np.sqrt(2 * N) * 1.5, (full.std(), proj.std())
    qf, qp = np.quantile(full, [0.25, 0.5, 0.75]), np.quantile(proj, [0.25, 0.5, 0.75])
    assert np.all(np.abs(qf - qp) <= 0.08 * full.std()), (qf, qp)
    # a different row: different distribution (the test has power)
    other = L.dccf_predict(ctx, m, L.rand_struct(seed=3, step=1),
                           torch.tensor([[8, 12]], dtype=torch.int64, device=dev()).repeat(N, 1), 0.0).double().cpu().numpy()
    assert abs(other.mean() - full.mean()) > 5 * se
    # with dropout the projected path still runs and stays finite
    pd_ = L.dccf_predict_projected(ctx, m, L.rand_struct(seed=3, step=2), X[:257], 0.2, Pf, Lt)
    assert bool(torch.isfinite(pd_).all())


@pytest.mark.parametrize('K,opt_name,F', [(2, 'adam', 96), (5, 'adam', 96), (7, 'adagrad', 96), (3, 'gd', 96),
                                          (4, 'gd', 97)])       # (F = 97: rows of dW are not 16-byte aligned)
def test_lazy_regularisation_equals_dense_pass(L, K, opt_name, F):
    """Windowed lazy regularisation (dccf_opt_t.lazy_*): 41 steps with the window cycling many times, a predict in the middle
    (flush), a tail batch of another size, steps that are not announced and announcements that are not kept — against the same calls with the dense pass
    (lazy_K = 0).  Rows no batch ever touched must be BIT-IDENTICAL (their updates are the same operations in the same order,
    applied K at a time); the rest agrees to the float-atomic tolerance of any two runs."""
    from dccf_amd.models import DCCF, FusedOptimizer
    U, I, D, B = 1501, 977, 64, 48
    g = torch.Generator(device='cuda').manual_seed(5)
    feat = torch.randn(I, F, generator=g, device='cuda') * 0.05
    expo = torch.randn(U, I, generator=g, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(9)
    nst = 41
    # users from the first 300 only, so that most user rows are never touched; items from everywhere
    full = torch.stack([torch.stack([torch.randint(0, 300, (2 * B,), generator=gen, device='cuda'),
                                     torch.randint(0, I, (2 * B,), generator=gen, device='cuda')], 1) for _ in range(nst)])
    tail = torch.stack([torch.randint(0, 300, (2 * 7,), generator=gen, device='cuda'),
                        torch.randint(0, I, (2 * 7,), generator=gen, device='cuda')], 1)
    y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')])
    res = []
    for lazy_K in (0, K):
        m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=10, attribute_num=2, std=0.1, label_min=0, label_max=1,
                 feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=1, random_seed=11,
                 model_path='/tmp/x.pt', feature_embedding=feat, expo_prob=expo)
        torch.manual_seed(3)
        m.apply(m.init_paras)
        m.optimizer = FusedOptimizer(m, opt_name, 0.01, 1e-3)
        m.lazy_K = lazy_K
        m.train()
        mid = None
        for k in range(nst):
            batch = {'X': full[k], 'Y': y, 'rank': 1, 'train': True, 'dropout': 0.2}
            # every 9th step unannounced; every 9th + 2 announces a batch that does NOT come (the optimizer launch has then
            # claimed and caught up the wrong rows for the next step: those claims must be forgotten, not shadow the real rows)
            nxt = full[k + 1] if k + 1 < nst and k % 9 != 4 else None
            if k % 9 == 6:
                nxt = full[(k + 5) % nst]
            m.train_step(batch, X_next=nxt)
            if k == 17:       # evaluation in the middle of an epoch: everything must be current for it
                m.eval()
                mid = m.predict({'X': full[0][:16].contiguous(), 'dropout': 0.0})['prediction'].clone()
                m.train()
            if k == 29:
                m.train_step({'X': tail, 'Y': torch.cat([y[:7], y[B:B + 7]]), 'rank': 1, 'train': True, 'dropout': 0.2})
        assert (m.optimizer.lazy is not None) == (lazy_K > 0)
        sd = m.state_dict()             # flushes
        assert m.optimizer.lazy is None or (not m.optimizer.lazy.dirty and int(m.optimizer.lazy.last.min()) == m.optimizer.t == nst + 1)
        res.append((sd, m.optimizer.s1.clone() if m.optimizer.s1 is not None else None, mid, m))
    (a, a1, amid, ma), (b, b1, bmid, mb) = res
    never_u = torch.ones(U, dtype=torch.bool, device='cuda')
    never_u[:300] = False
    assert torch.equal(a['uid_embeddings.weight'][never_u], b['uid_embeddings.weight'][never_u])
    if a1 is not None:
        assert torch.equal(a1[:U * D].view(U, D)[never_u], b1[:U * D].view(U, D)[never_u])
    lr, steps = 0.01, nst + 1
    for k in a:
        d = (a[k] - b[k]).abs()
        if opt_name == 'gd':
            assert float(d.max()) <= 2e-6, (k, float(d.max()))
        else:       # Adam / Adagrad turn the last-bit differences of float-atomic sums into fractions of lr on a few elements
            assert float(d.max()) <= steps * lr and float((d > steps * 5e-3 * lr).float().mean()) <= 0.02, (k, float(d.max()))
    close(bmid, amid.cpu().numpy(), 2e-3 if opt_name != 'gd' else 1e-5, 1e-5, 'mid-epoch evaluation')


def test_reference_checkpoint_loads(L):
    """A .pt written by the REFERENCE's save_model (tests/golden/dccf_d24_f100_l3_adagrad.pt, src/models/BaseModel.py:224-236)
    loads into this build's DCCF (same state_dict keys and shapes: 3 mlp layers, D = 24) and predicts the reference's own
    evaluation output on the captured draws; and what this build saves, torch loads back with the same keys."""
    import os
    from conftest import GOLDEN
    from dccf_amd.models import DCCF
    g = load_golden('dccf_d24_f100_l3_adagrad')
    m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=int(g['S']), attribute_num=int(g['A']), std=float(g['std']),
             label_min=0, label_max=1, feature_num=0, user_num=int(g['U']), item_num=int(g['I']), u_vector_size=int(g['D']),
             i_vector_size=int(g['D']), n_layers=int(g['n_layers']), random_seed=1, model_path=os.path.join(GOLDEN, 'dccf_d24_f100_l3_adagrad.pt'),
             feature_embedding=T(g['feat']), expo_prob=T(g['expo']))
    m.load_model()
    out = m.predict({'X': T(g['eval/X']), 'dropout': 0.0,
                     'inject': {'sample_item': T(g['eval/sample_item']), 'noise': T(g['eval/noise'])}})['prediction']
    close(out, g['eval/prediction'], FWD_RTOL, FWD_ATOL, 'prediction from the reference checkpoint')
    path = '/tmp/dccf_mine_l3.pt'
    m.save_model(path)
    sd = torch.load(path, map_location='cpu')
    assert list(sd.keys()) == pkeys(g)
    last = 's%d/after/' % (int(g['steps']) - 1)
    for k in sd:
        assert np.array_equal(sd[k].numpy(), g[last + k])


def test_init_paras_distribution(L):
    """BaseModel.init_paras through model.apply (src/models/BaseModel.py:130-142, src/main.py:150): every Embedding weight, every
    Linear weight AND bias ~ N(0, 0.01); a bare Parameter (BiasedMF.global_bias = 0.1, src/models/BiasedMF.py:14) is not a
    module and keeps its value.  Distribution, not stream: mean / std / excess kurtosis within sampling error."""
    from dccf_amd.models import DCCF, BiasedMF
    feat = torch.randn(300, 48, device='cuda')
    expo = torch.randn(400, 300, device='cuda')
    m = DCCF(path=None, dataset=None, sentence_model=None, sample_num=4, attribute_num=2, std=0.1, label_min=0, label_max=1,
             feature_num=0, user_num=400, item_num=300, u_vector_size=64, i_vector_size=64, n_layers=3, random_seed=5,
             model_path='/tmp/x.pt', feature_embedding=feat, expo_prob=expo)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    m.apply(m.init_paras)
    P = dict(m.named_parameters())
    assert set(P) == {'uid_embeddings.weight', 'iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias', 'mlp.1.weight',
                      'mlp.1.bias', 'mlp.2.weight', 'mlp.2.bias'}
    for k, v in P.items():
        x = v.detach().double().flatten()
        n = x.numel()
        assert not torch.equal(v.detach(), before[k]), k
        assert abs(float(x.mean())) < 5 * 0.01 / n ** 0.5 + 1e-12, k
        if n >= 1000:
            assert float(x.std()) == pytest.approx(0.01, rel=5 * (0.5 / n) ** 0.5 + 1e-3), k
            kurt = float(((x - x.mean()) ** 4).mean() / x.var() ** 2) - 3.0
            assert abs(kurt) < 6 * (24.0 / n) ** 0.5 + 0.02, k
        else:                      # the 64-element biases: inside 5 sigma, not the torch Linear default U(-1/8, 1/8)
            assert float(x.abs().max()) < 0.05 and float(x.std()) == pytest.approx(0.01, rel=0.5), k
    b = BiasedMF(label_min=0, label_max=1, feature_num=0, user_num=400, item_num=300, u_vector_size=16, i_vector_size=16,
                 random_seed=5, model_path='/tmp/x.pt')
    b.apply(b.init_paras)
    Pb = dict(b.named_parameters())
    assert float(Pb['global_bias']) == pytest.approx(0.1)
    for k in ('user_bias.weight', 'item_bias.weight', 'uid_embeddings.weight', 'iid_embeddings.weight'):
        x = Pb[k].detach().double().flatten()
        assert float(x.std()) == pytest.approx(0.01, rel=0.25) and abs(float(x.mean())) < 0.003, k


@pytest.mark.parametrize('kind,opt_name,rank,D', [('BiasedMF', 'adam', 1, 64), ('RecModel', 'adagrad', 1, 64), ('IPSBiasedMF', 'gd', 0, 64),
                                                  ('IPSBiasedMF', 'adam', 1, 64),
                                                  # round 3: embedding sizes that are not a kernel tile take the lazy path too
                                                  ('IPSBiasedMF', 'adam', 1, 24), ('BiasedMF', 'adagrad', 1, 100), ('RecModel', 'gd', 1, 160)])
def test_mf_lazy_train_step_equals_dense_step(L, kind, opt_name, rank, D):
    """MF family: train_step (mf_train_step: catch-up of the batch's rows, forward + backward, one lazy optimizer launch; window
    K = 4 so that it cycles several times) against forward + the dense optimizer step (src/runners/BaseRunner.py:172-188), a
    mid-run evaluation (flush) included: rows no batch touched are bit-identical, the others agree to the float-atomic tolerance,
    predictions of the steps agree, nothing is left in the gradient buffer."""
    from dccf_amd import models
    U, I, B, steps = 2203, 1877, 48, 23
    runs, preds, seen_u = [], [], []
    for lazy in (True, False):
        cls = models.BiasedMF if kind != 'RecModel' else models.RecModel
        m = cls(label_min=0, label_max=1, feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, random_seed=4,
                model_path='/tmp/mf.pt')
        if kind == 'IPSBiasedMF':
            m.kind, m.M = 'IPSBiasedMF', 0.1
            m.propensity = torch.rand(I, generator=torch.Generator(device='cuda').manual_seed(9), device='cuda')
        torch.manual_seed(1)
        m.apply(m.init_paras)
        m.lazy_K = 4 if lazy else 0
        m.optimizer = models.FusedOptimizer(m, opt_name, 0.01, 1e-3)
        m.train()
        gen = torch.Generator(device='cuda').manual_seed(2)
        y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')]) if rank == 1 else \
            torch.rand(2 * B, generator=torch.Generator(device='cuda').manual_seed(3), device='cuda')
        ps = []
        for k in range(steps):
            u = torch.randint(0, U // 2, (B,), generator=gen, device='cuda')           # the upper half of the users: never touched
            X = torch.stack([torch.cat([u, u]), torch.randint(0, I, (2 * B,), generator=gen, device='cuda')], 1)
            out = m.train_step({'X': X, 'Y': y, 'rank': rank, 'train': True, 'dropout': 0.0})
            ps.append(out['prediction'].clone())
            if k == 9:                       # an evaluation in the middle of the epoch reads every row
                m.eval()
                ps.append(m.predict({'X': X})['prediction'].clone())
                m.train()
        assert (m.optimizer.lazy is not None) == lazy and m.optimizer.t == steps
        sd = m.state_dict()                 # (flushes)
        torch.cuda.synchronize()
        assert float(m.flat_g.abs().max()) == 0.0
        runs.append(m.flat_p.clone())
        preds.append(torch.stack([p for p in ps]))
        never = m.params['uid_embeddings.weight'][U // 2:].clone()
        seen_u.append(never)
    assert torch.equal(seen_u[0], seen_u[1])                                   # untouched rows: bit for bit
    assert float(seen_u[0].abs().max()) > 0 and not torch.equal(seen_u[0][0], torch.zeros(D, device='cuda'))
    d = (runs[0] - runs[1]).abs()
    tol = steps * 0.01
    assert float(d.max()) <= tol and int((d > steps * STEP_FRAC * 0.01).sum()) <= 4 * D + 8
    assert float((preds[0] - preds[1]).abs().max()) <= 5e-3 * max(1.0, float(preds[1].abs().max()))
