# coding=utf-8
"""Generate golden vectors by RUNNING THE UNMODIFIED REFERENCE on CPU (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (read-only, /root/reference/src) is imported in this process with process-local shims only
(SURVEY.md §8c): a stub ``pymining`` module, the removed numpy aliases, and CUDA entry points redirected
to CPU.  Randomness inside ``DCCF.predict`` (``src/models/DCCF.py:72,87,94``) is *captured*: the candidate
items drawn by ``torch.randint``, the Gaussian feature noise drawn by ``torch.cuda.FloatTensor(..).normal_``
and the dropout keep-mask are recorded next to the outputs so that the oracle (oracle/) and the HIP path can
be fed the same values.  Nothing from the reference is copied: the .npz files hold inputs and outputs only.
This script never runs on the GPU box (it needs /root/reference).
"""
import os
import sys
import types
import tempfile
import shutil
import zlib

import numpy as np
import torch

REF_SRC = '/root/reference/src'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


# ----------------------------------------------------------------------------- shims
class Capture(object):
    """Records / injects the random draws of DCCF.predict."""

    def __init__(self):
        self.reset()

    enabled = True            # False for the end-to-end runs: nothing is recorded (a 5k x 5k evaluation draws ~1 GB per batch)

    def reset(self):
        self.sample_item = []
        self.noise = []
        self.masks = []


CAP = Capture()


def install_shims():
    pm = types.ModuleType('pymining')
    for sub in ('itemmining', 'assocrules', 'perftesting'):
        m = types.ModuleType('pymining.' + sub)
        setattr(pm, sub, m)
        sys.modules['pymining.' + sub] = m
    sys.modules['pymining'] = pm
    if not hasattr(np, 'asfarray'):
        np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)
    if not hasattr(np, 'float'):
        np.float = float
    if not hasattr(np, 'int'):
        np.int = int

    torch.cuda.current_device = lambda: 0
    torch.cuda.manual_seed = lambda s: None
    _orig_to = torch.Tensor.to

    def _to(self, *args, **kwargs):
        if args and isinstance(args[0], str) and args[0].startswith('cuda'):
            return self
        return _orig_to(self, *args, **kwargs)

    torch.Tensor.to = _to

    _orig_randint = torch.randint

    def _randint(*args, **kwargs):
        t = _orig_randint(*args, **kwargs)
        if CAP.enabled:
            CAP.sample_item.append(t.clone())
        return t

    torch.randint = _randint

    class _NoiseTensor(object):
        def __init__(self, shape):
            self.shape = shape

        def normal_(self, mean=0.0, std=1.0):
            t = torch.empty(self.shape, dtype=torch.float32).normal_(mean=mean, std=std)
            if CAP.enabled:
                CAP.noise.append(t.clone())
            return t

    torch.cuda.FloatTensor = lambda shape: _NoiseTensor(shape)

    class _Dropout(torch.nn.Module):
        """Same arithmetic as torch's dropout (x * bernoulli(1-p) / (1-p)), mask recorded."""

        def __init__(self, p=0.5, inplace=False):
            super().__init__()
            self.p = p

        def forward(self, x):
            if self.p == 0.0:
                if CAP.enabled:
                    CAP.masks.append(torch.ones_like(x))
                return x
            keep = torch.empty_like(x).bernoulli_(1.0 - self.p)
            if CAP.enabled:
                CAP.masks.append(keep.clone())
            return x * keep.div(1.0 - self.p)

    torch.nn.Dropout = _Dropout
    sys.path.insert(0, REF_SRC)


# ----------------------------------------------------------------------------- helpers
def t2n(t):
    return t.detach().cpu().numpy().copy()


def params_of(model):
    return {k: t2n(v) for k, v in model.state_dict().items()}


def grads_of(model):
    return {k: t2n(p.grad) for k, p in model.named_parameters()}


def make_pair_batch(rng, user_num, item_num, pairs):
    """Layout of DataProcessor._get_feed_dict_rk (src/data_processor/DataProcessor.py:160-207):
    X = [pos rows ; neg rows], row k and row B+k carry the same uid."""
    u = rng.randint(0, user_num, size=pairs)
    ip = rng.randint(0, item_num, size=pairs)
    ineg = rng.randint(0, item_num, size=pairs)
    X = np.concatenate([np.stack([u, ip], 1), np.stack([u, ineg], 1)], 0).astype(np.int64)
    Y = np.concatenate([np.ones(pairs, np.float32), np.zeros(pairs, np.float32)])
    return X, Y


def feed(X, Y, rank, dropout, train=True):
    return {'X': torch.from_numpy(X), 'Y': torch.from_numpy(Y), 'rank': rank, 'train': train,
            'dropout': dropout, 'sample_id': np.arange(len(Y))}


def fit_like_reference(model, runner, batch):
    """Body of BaseRunner.fit's loop (src/runners/BaseRunner.py:178-187) with grads recorded around the clip."""
    if model.optimizer is None:
        model.optimizer = runner._build_optimizer(model)
    model.train()
    model.optimizer.zero_grad()
    CAP.reset()
    out = model(batch)
    loss_only = out['loss']
    for p in model.parameters():
        p.grad = None
    loss_only.backward(retain_graph=True)
    g_loss = grads_of(model)
    model.optimizer.zero_grad()
    l2v = model.l2()
    loss = out['loss'] + l2v * runner.l2_weight
    loss.backward()
    g_pre = grads_of(model)
    torch.nn.utils.clip_grad_value_(model.parameters(), 50)
    g_post = grads_of(model)
    model.optimizer.step()
    return out, t2n(l2v), g_loss, g_pre, g_post


# ----------------------------------------------------------------------------- G1: DCCF
def gen_dccf(outdir, cases=None):
    from models.DCCF import DCCF
    from runners.BaseRunner import BaseRunner
    cases = cases or [
        # name, U, I, D, F, pairs, S, A, std, dropout, optimizer, lr, l2, steps, rank, init_scale
        ('dccf_d16_f32_adam', 50, 40, 16, 32, 4, 10, 2, 0.1, 0.2, 'Adam', 0.001, 1e-4, 3, 1, 1.0),
        ('dccf_d64_f768_adam', 50, 40, 64, 768, 4, 10, 2, 0.1, 0.2, 'Adam', 0.001, 1e-4, 3, 1, 1.0),
        ('dccf_d64_f32_nodrop_gd', 50, 40, 64, 32, 6, 10, 2, 0.1, 0.0, 'GD', 0.01, 1e-4, 3, 1, 1.0),
        ('dccf_d32_f160_adagrad', 30, 70, 32, 160, 5, 4, 3, 0.3, 0.5, 'Adagrad', 0.01, 1e-3, 3, 1, 30.0),
        ('dccf_d128_f768_adam', 20, 25, 128, 768, 2, 10, 2, 0.1, 0.2, 'Adam', 0.001, 1e-4, 2, 1, 10.0),
        ('dccf_d64_f768_mse', 50, 40, 64, 768, 4, 10, 2, 0.1, 0.2, 'Adam', 0.001, 1e-4, 2, 0, 20.0),
    ]
    for case in cases:
        (name, U, I, D, F, pairs, S, A, std, p, opt, lr, l2, steps, rank, scale) = case[:16]
        n_layers = case[16] if len(case) > 16 else 1          # --n_layers (src/models/DMF.py:14)
        tmp = tempfile.mkdtemp()
        try:
            rng = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
            feat = (rng.randn(I, F) * 0.05 * scale).astype(np.float32)
            expo = rng.randn(U, I).astype(np.float32)
            np.save(os.path.join(tmp, 'toy_sm.npy'), feat)
            np.save(os.path.join(tmp, 'toy.ips_expo_prob.npy'), expo)
            model = DCCF(path=tmp, dataset='toy', sentence_model='sm', sample_num=S, attribute_num=A, std=std,
                         label_min=0, label_max=1, feature_num=0, user_num=U, item_num=I, u_vector_size=D,
                         i_vector_size=D, n_layers=n_layers, random_seed=2019, model_path=os.path.join(tmp, 'm.pt'))
            model.apply(model.init_paras)
            if scale != 1.0:   # larger weights: exercises relu / softmax / sigmoid away from the linear regime
                with torch.no_grad():
                    for q in model.parameters():
                        q.mul_(scale)
            runner = BaseRunner(optimizer=opt, learning_rate=lr, epoch=1, batch_size=pairs, dropout=p, l2=l2,
                                metrics='ndcg@5')
            rec = {'U': U, 'I': I, 'D': D, 'F': F, 'pairs': pairs, 'S': S, 'A': A, 'std': std, 'dropout': p,
                   'lr': lr, 'l2': l2, 'steps': steps, 'rank': rank, 'optimizer': opt,
                   'feat': feat, 'expo': expo}
            if n_layers != 1:
                rec['n_layers'] = n_layers
            for k, v in params_of(model).items():
                rec['init/' + k] = v
            for s in range(steps):
                if rank == 1:
                    X, Y = make_pair_batch(rng, U, I, pairs)
                else:
                    X = np.stack([rng.randint(0, U, size=2 * pairs), rng.randint(0, I, size=2 * pairs)], 1).astype(np.int64)
                    Y = rng.randint(0, 2, size=2 * pairs).astype(np.float32)
                out, l2v, g_loss, g_pre, g_post = fit_like_reference(model, runner, feed(X, Y, rank, p))
                pre = 's%d/' % s
                rec[pre + 'X'], rec[pre + 'Y'] = X, Y
                rec[pre + 'sample_item'] = t2n(CAP.sample_item[0])
                rec[pre + 'noise'] = t2n(CAP.noise[0])
                if n_layers == 1:
                    rec[pre + 'mask'] = t2n(CAP.masks[0]).astype(np.uint8)
                else:          # one keep mask per mlp layer (src/models/DCCF.py:91-94), layer-major
                    assert len(CAP.masks) == n_layers
                    rec[pre + 'mask'] = np.stack([t2n(mk).astype(np.uint8) for mk in CAP.masks])
                rec[pre + 'prediction'] = t2n(out['prediction'])
                rec[pre + 'loss'] = t2n(out['loss'])
                rec[pre + 'l2'] = l2v
                for k in g_pre:
                    rec[pre + 'gloss/' + k] = g_loss[k]
                    rec[pre + 'gpre/' + k] = g_pre[k]
                    rec[pre + 'gpost/' + k] = g_post[k]
                for k, v in params_of(model).items():
                    rec[pre + 'after/' + k] = v
            # eval-mode predict (dropout 0.0, src/runners/BaseRunner.py:131,150)
            model.eval()
            Xe = np.stack([rng.randint(0, U, size=11), rng.randint(0, I, size=11)], 1).astype(np.int64)
            CAP.reset()
            with torch.no_grad():
                pe = model.predict(feed(Xe, np.zeros(11, np.float32), 1, 0.0, train=False))['prediction']
            rec['eval/X'] = Xe
            rec['eval/sample_item'] = t2n(CAP.sample_item[0])
            rec['eval/noise'] = t2n(CAP.noise[0])
            rec['eval/prediction'] = t2n(pe)
            np.savez_compressed(os.path.join(outdir, name + '.npz'), **rec)
            if len(case) > 17 and case[17]:
                # a checkpoint written by the reference's own save_model (src/models/BaseModel.py:224-236) after the steps above:
                # the .pt interchange fixture (tests/test_hip_parity.py::test_reference_checkpoint_loads)
                model.model_path = os.path.join(outdir, name + '.pt')
                model.save_model()
            print('wrote', name, 'loss', [float(rec['s%d/loss' % s]) for s in range(steps)])
        finally:
            shutil.rmtree(tmp)


def gen_dccf_ext(outdir):
    """Round-2 cases, generated AFTER everything else with their own torch seed so that the round-1 fixtures regenerate bit
    for bit: --n_layers > 1 (src/models/DMF.py:14, src/models/DCCF.py:61-62,91-94), embedding widths that are not a kernel
    tile (src/models/RecModel.py:17-27 accepts any) and feature files wider than 896 (src/models/DCCF.py:59)."""
    torch.manual_seed(2020)
    gen_dccf(outdir, cases=[
        # name, U, I, D, F, pairs, S, A, std, dropout, optimizer, lr, l2, steps, rank, init_scale, n_layers
        ('dccf_d64_f768_l2_adam', 50, 40, 64, 768, 4, 10, 2, 0.1, 0.2, 'Adam', 0.001, 1e-4, 3, 1, 8.0, 2),
        ('dccf_d24_f100_l3_adagrad', 40, 30, 24, 100, 5, 6, 2, 0.2, 0.3, 'Adagrad', 0.01, 1e-3, 2, 1, 25.0, 3, True),
        ('dccf_d48_f1024_adam', 30, 35, 48, 1024, 3, 10, 2, 0.1, 0.2, 'Adam', 0.001, 1e-4, 2, 1, 10.0, 1),
        ('dccf_d100_f800_gd', 20, 25, 100, 800, 2, 10, 2, 0.1, 0.2, 'GD', 0.01, 1e-4, 2, 0, 10.0, 1),
        ('dccf_d128_f32_l2_mse', 20, 25, 128, 32, 3, 4, 3, 0.1, 0.2, 'Adam', 0.001, 1e-4, 2, 0, 20.0, 2),
    ])


def gen_dccf_wide(outdir):
    """Round-3 cases: embedding sizes above 128 (src/models/RecModel.py:17-27 accepts any u_vector_size == i_vector_size), with
    their own torch seed so that the earlier fixtures regenerate bit for bit."""
    torch.manual_seed(2021)
    gen_dccf(outdir, cases=[
        # name, U, I, D, F, pairs, S, A, std, dropout, optimizer, lr, l2, steps, rank, init_scale, n_layers
        ('dccf_d192_f768_adam', 20, 25, 192, 768, 2, 10, 2, 0.1, 0.2, 'Adam', 0.001, 1e-4, 2, 1, 10.0, 1),
        ('dccf_d256_f96_adagrad', 16, 20, 256, 96, 3, 4, 2, 0.2, 0.3, 'Adagrad', 0.01, 1e-3, 2, 1, 10.0, 1),
        ('dccf_d160_f1000_gd_mse', 18, 22, 160, 1000, 2, 6, 2, 0.1, 0.2, 'GD', 0.01, 1e-4, 2, 0, 10.0, 1),
    ])


# ----------------------------------------------------------------------------- G2: MF family
def gen_mf(outdir):
    from models.RecModel import RecModel
    from models.BiasedMF import BiasedMF
    from models.IPSBiasedMF import IPSBiasedMF
    from runners.BaseRunner import BaseRunner
    U, I, D, pairs = 37, 29, 64, 9
    for name, cls in (('mf_recmodel', RecModel), ('mf_biasedmf', BiasedMF), ('mf_ipsbiasedmf', IPSBiasedMF)):
        tmp = tempfile.mkdtemp()
        try:
            rng = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
            kw = dict(label_min=0, label_max=1, feature_num=0, user_num=U, item_num=I, u_vector_size=D,
                      i_vector_size=D, random_seed=2019, model_path=os.path.join(tmp, 'm.pt'))
            rec = {'U': U, 'I': I, 'D': D, 'pairs': pairs, 'lr': 0.01, 'l2': 1e-4, 'M': 0.1}
            if cls is IPSBiasedMF:
                prop = rng.rand(I).astype(np.float32)   # some below M=0.1, some above
                np.save(os.path.join(tmp, 'toy.propensity.npy'), prop)
                rec['propensity'] = prop
                model = cls(path=tmp, dataset='toy', M=0.1, **kw)
            else:
                model = cls(**kw)
            model.apply(model.init_paras)
            with torch.no_grad():
                for q in model.parameters():
                    if q.dim() > 0:
                        q.mul_(30.0)
            runner = BaseRunner(optimizer='Adam', learning_rate=0.01, epoch=1, batch_size=pairs, dropout=0.2, l2=1e-4,
                                metrics='ndcg@5')
            for k, v in params_of(model).items():
                rec['init/' + k] = v
            for s in range(2):
                X, Y = make_pair_batch(rng, U, I, pairs)
                out, l2v, g_loss, g_pre, g_post = fit_like_reference(model, runner, feed(X, Y, 1, 0.2))
                pre = 's%d/' % s
                rec[pre + 'X'], rec[pre + 'Y'] = X, Y
                rec[pre + 'prediction'] = t2n(out['prediction'])
                rec[pre + 'loss'] = t2n(out['loss'])
                rec[pre + 'l2'] = l2v
                for k in g_pre:
                    rec[pre + 'gloss/' + k] = g_loss[k]
                    rec[pre + 'gpre/' + k] = g_pre[k]
                    rec[pre + 'gpost/' + k] = g_post[k]
                for k, v in params_of(model).items():
                    rec[pre + 'after/' + k] = v
            # the full U x I matrix ("save the full predicted user-item matrix", README.md:28-30), by looping predict
            model.eval()
            uu, ii = np.meshgrid(np.arange(U), np.arange(I), indexing='ij')
            Xf = np.stack([uu.reshape(-1), ii.reshape(-1)], 1).astype(np.int64)
            with torch.no_grad():
                full = model.predict(feed(Xf, np.zeros(len(Xf), np.float32), 1, 0.0, train=False))['prediction']
            rec['full'] = t2n(full).reshape(U, I)
            np.savez_compressed(os.path.join(outdir, name + '.npz'), **rec)
            print('wrote', name)
        finally:
            shutil.rmtree(tmp)


# ----------------------------------------------------------------------------- G-opt: l2 + clip + optimizers
def gen_opt(outdir):
    """Dense regularised step: grad = sparse + l2w*2p (autograd of BaseModel.l2, src/models/BaseModel.py:179-187),
    clip_grad_value_(50) (src/runners/BaseRunner.py:185), torch.optim.{SGD,Adagrad,Adam}(weight_decay=l2)
    (src/runners/BaseRunner.py:92,96,100).  Sparse grads are scaled so the clip triggers."""
    from models.RecModel import RecModel
    from runners.BaseRunner import BaseRunner
    for opt in ('GD', 'Adagrad', 'Adam'):
        rng = np.random.RandomState(11)
        model = RecModel(label_min=0, label_max=1, feature_num=0, user_num=23, item_num=17, u_vector_size=8,
                         i_vector_size=8, random_seed=2019, model_path='/tmp/none.pt')
        model.apply(model.init_paras)
        with torch.no_grad():
            for q in model.parameters():
                q.mul_(100.0)
        runner = BaseRunner(optimizer=opt, learning_rate=0.05, epoch=1, batch_size=4, dropout=0.0, l2=1e-2,
                            metrics='ndcg@5')
        model.optimizer = runner._build_optimizer(model)
        rec = {'lr': 0.05, 'l2': 1e-2, 'optimizer': opt, 'steps': 4}
        for k, v in params_of(model).items():
            rec['init/' + k] = v
        for s in range(4):
            model.optimizer.zero_grad()
            sparse = {}
            for k, q in model.named_parameters():
                g = (rng.randn(*q.shape) * 40.0).astype(np.float32)
                g[rng.rand(*q.shape) < 0.5] = 0.0
                sparse[k] = g
                q.grad = torch.from_numpy(g.copy())
            (model.l2() * runner.l2_weight).backward()
            pre = 's%d/' % s
            for k, q in model.named_parameters():
                rec[pre + 'sparse/' + k] = sparse[k]
                rec[pre + 'gpre/' + k] = t2n(q.grad)
            torch.nn.utils.clip_grad_value_(model.parameters(), 50)
            for k, q in model.named_parameters():
                rec[pre + 'gpost/' + k] = t2n(q.grad)
            model.optimizer.step()
            for k, v in params_of(model).items():
                rec[pre + 'after/' + k] = v
        np.savez_compressed(os.path.join(outdir, 'opt_%s.npz' % opt.lower()), **rec)
        print('wrote opt', opt)


# ----------------------------------------------------------------------------- G3: data processor
def gen_batches(outdir):
    """DataLoader + DataProcessor on a toy set: train negatives / feed dicts for a fixed numpy seed, eval dicts."""
    from data_loaders.DataLoader import DataLoader
    from data_processor.DataProcessor import DataProcessor
    from models.RecModel import RecModel
    from utils import utils as rutils
    from dccf_amd import synth
    tmp = tempfile.mkdtemp()
    try:
        synth.write_dataset(tmp, 'toyds', user_num=60, item_num=45, n_draws=700, feat_dim=8, seed=3)
        dl = DataLoader(path=tmp, dataset='toyds', label='label', sep=',')
        model = RecModel(label_min=0, label_max=1, feature_num=0, user_num=dl.user_num, item_num=dl.item_num,
                         u_vector_size=4, i_vector_size=4, random_seed=2019, model_path='/tmp/none.pt')
        dl.drop_neg()
        dp = DataProcessor(dl, model, rank=1, test_neg_n=10)
        rec = {'user_num': dl.user_num, 'item_num': dl.item_num, 'test_neg_n': 10, 'batch_size': 32, 'np_seed': 2019}
        for nm, df in (('train', dl.train_df), ('validation', dl.validation_df), ('test', dl.test_df)):
            rec['df/' + nm] = df[['uid', 'iid', 'label', 'time']].values.astype(np.int64)
        np.random.seed(2019)
        # order of main.py: test data first ("Test Before Training", src/main.py:181), then train(): train(-1), validation, test
        test = dp.get_test_data()
        train = dp.get_train_data(epoch=-1)
        val = dp.get_validation_data()
        for nm, d in (('test', test), ('validation', val)):
            for k in ('uid', 'iid', 'Y', 'X', 'sample_id'):
                rec['%s/%s' % (nm, k)] = np.asarray(d[k])
        for ep in range(2):
            tr = dp.get_train_data(epoch=ep)
            for k in ('uid', 'iid', 'Y', 'X', 'sample_id'):
                rec['train_ep%d/%s' % (ep, k)] = np.asarray(tr[k]).copy()
            batches = dp.prepare_batches(tr, 32, train=True)
            rec['train_ep%d/n_batches' % ep] = len(batches)
            rec['train_ep%d/batch_X' % ep] = np.concatenate([t2n(b['X']) for b in batches], 0)
            rec['train_ep%d/batch_Y' % ep] = np.concatenate([t2n(b['Y']) for b in batches], 0)
            rec['train_ep%d/batch_sample_id' % ep] = np.concatenate([b['sample_id'] for b in batches], 0)
            rec['train_ep%d/batch_sizes' % ep] = np.array([b['real_batch_size'] for b in batches])
        vb = dp.prepare_batches(val, 64, train=False)
        rec['validation/batch_X'] = np.concatenate([t2n(b['X']) for b in vb], 0)
        rec['validation/batch_sizes'] = np.array([len(b['Y']) for b in vb])
        np.savez_compressed(os.path.join(outdir, 'batches.npz'), **rec)
        print('wrote batches: train', len(train['Y']), 'val', len(val['Y']), 'test', len(test['Y']))
    finally:
        shutil.rmtree(tmp)


# ----------------------------------------------------------------------------- G4: metrics
def gen_metrics(outdir):
    from models.BaseModel import BaseModel
    from utils import rank_metrics as rm
    rng = np.random.RandomState(5)
    n_users, per = 40, 23
    uid = np.repeat(np.arange(n_users), per)
    l = (rng.rand(n_users * per) < 0.15).astype(np.float32)
    l[::per] = 1.0   # every user has at least one positive (recall divides by sum(l))
    p = rng.randn(n_users * per).astype(np.float32)   # tie-free with probability 1
    perm = rng.permutation(len(uid))
    uid, l, p = uid[perm], l[perm], p[perm]
    metrics = ['ndcg@1', 'ndcg@5', 'ndcg@10', 'hit@5', 'precision@5', 'recall@5', 'recall@10', 'f1@5', 'rmse', 'mae']
    vals = BaseModel.evaluate_method(p, {'uid': uid, 'Y': l}, metrics)
    rec = {'uid': uid, 'Y': l, 'p': p, 'metrics': np.array(metrics), 'values': np.array(vals, dtype=np.float64)}
    r = [3, 2, 3, 0, 0, 1, 2, 2, 3, 0]
    rec['doc/dcg'] = np.array([rm.dcg_at_k(r, 1), rm.dcg_at_k(r, 1, method=1), rm.dcg_at_k(r, 2), rm.dcg_at_k(r, 2, method=1),
                               rm.dcg_at_k(r, 10), rm.dcg_at_k(r, 11)])
    rec['doc/ndcg'] = np.array([rm.ndcg_at_k(r, 1), rm.ndcg_at_k([2, 1, 2, 0], 4), rm.ndcg_at_k([2, 1, 2, 0], 4, method=1),
                                rm.ndcg_at_k([0], 1), rm.ndcg_at_k([1], 2)])
    rec['doc/precision'] = np.array([rm.precision_at_k([0, 0, 1], 1), rm.precision_at_k([0, 0, 1], 2), rm.precision_at_k([0, 0, 1], 3)])
    np.savez_compressed(os.path.join(outdir, 'metrics.npz'), **rec)
    print('wrote metrics', dict(zip(metrics, vals)))


# ----------------------------------------------------------------------------- G5: end to end
E2E_CONFIGS = {
    # small set the first e2e golden was captured on (tests/golden/e2e.npz)
    'small': dict(user_num=800, item_num=600, n_draws=16000, feat_dim=64, data_seed=11, epochs=4, D=32, test_neg_n=100,
                  lr=0.001, batch_size=128),
    # BASELINE.json configs[0] / SURVEY.md §8d C1: 5k users x 5k items, D=16, F=768 (tests/golden/e2e_c1.npz)
    'c1': dict(user_num=5000, item_num=5000, n_draws=200000, feat_dim=768, data_seed=7, epochs=3, D=16, test_neg_n=100,
               lr=0.001, batch_size=128),
}
E2E = dict(E2E_CONFIGS[os.environ.get('E2E_CFG', 'small')])
E2E['seeds'] = [int(x) for x in os.environ.get('E2E_SEEDS', '2019,2020,2021,2022,2023').split(',')]
if 'E2E_THREADS' in os.environ:
    torch.set_num_threads(int(os.environ['E2E_THREADS']))


def gen_e2e(outdir):
    """The reference's own main.py (src/main.py) on a small synthetic dataset: per-epoch validation / test metrics for
    several seeds — the statistical target of tests/test_e2e_gpu.py (evaluation is stochastic)."""
    import re
    import io
    import logging
    import contextlib
    from dccf_amd import synth
    import main as ref_main
    c = E2E
    CAP.enabled = False
    rec = {k: np.array(v) for k, v in c.items()}
    cwd = os.getcwd()
    done = []
    for seed in c['seeds']:
        tmp = tempfile.mkdtemp()
        try:
            os.makedirs(os.path.join(tmp, 'src'))
            os.makedirs(os.path.join(tmp, 'result'))
            synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', c['user_num'], c['item_num'], c['n_draws'],
                                feat_dim=c['feat_dim'], seed=c['data_seed'])
            os.chdir(os.path.join(tmp, 'src'))
            sys.argv = ['main.py', '--rank', '1', '--model_name', 'DCCF', '--optimizer', 'Adam', '--lr', str(c['lr']),
                        '--dataset', 'toy', '--path', '../dataset/', '--metric', 'ndcg@5,recall@5,precision@5', '--gpu', '',
                        '--epoch', str(c['epochs']), '--test_neg_n', str(c['test_neg_n']), '--u_vector_size', str(c['D']),
                        '--i_vector_size', str(c['D']), '--random_seed', str(seed), '--batch_size', str(c['batch_size']),
                        '--check_epoch', '0']
            ref_main.main()
            logf = [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(tmp, 'log')) for f in fs][0]
            txt = open(logf).read()
            ep = re.findall(r'Epoch\s+(\d+) \[[\d.]+ s\]\s+train= ([\d.,-]+) validation= ([\d.,-]+) test= ([\d.,-]+)', txt)
            tms = re.findall(r'Epoch\s+\d+ \[([\d.]+) s\].*?\[([\d.]+) s\]', txt)
            rec['seed%d/fit_eval_seconds' % seed] = np.array([[float(a), float(b)] for a, b in tms])
            rec['threads'] = np.array(torch.get_num_threads())
            init = re.search(r'Init: \s+train= ([\d.,-]+) validation= ([\d.,-]+) test= ([\d.,-]+)', txt)
            rec['seed%d/init_valid' % seed] = np.array([float(x) for x in init.group(2).split(',')])
            rec['seed%d/init_test' % seed] = np.array([float(x) for x in init.group(3).split(',')])
            rec['seed%d/valid' % seed] = np.array([[float(x) for x in e[2].split(',')] for e in ep])
            rec['seed%d/test' % seed] = np.array([[float(x) for x in e[3].split(',')] for e in ep])
            print('seed', seed, 'valid ndcg@5 per epoch', rec['seed%d/valid' % seed][:, 0], flush=True)
            done.append(seed)
            rec['seeds'] = np.array(done)          # saved after every seed: a run takes minutes per seed
            np.savez_compressed(os.path.join(outdir, os.environ.get('E2E_OUT', 'e2e.npz')), **rec)
        finally:
            os.chdir(cwd)
            for h in logging.root.handlers[:]:
                logging.root.removeHandler(h)
            shutil.rmtree(tmp)


def gen_e2e_init(outdir):
    """The UNTRAINED model's validation metrics by the reference's own classes (what its `Init:` log line reports,
    src/runners/BaseRunner.py:229-237), many seeds, without the rest of main.py's run: DataLoader -> DCCF -> init_paras ->
    drop_neg -> DataProcessor -> BaseRunner.evaluate(validation), constructed as src/main.py:100-174 does.  One evaluation pass
    per seed (~2 minutes on config 1's shape) instead of a whole run (~25): the statistical target for the evaluation path alone
    (tests/test_e2e_gpu.py), where round 2 saw a +0.0013 offset at 2.6 standard errors with 11 reference seeds."""
    import logging
    from dccf_amd import synth
    from data_loaders.DataLoader import DataLoader
    from data_processor.DataProcessor import DataProcessor
    from runners.BaseRunner import BaseRunner
    from models.DCCF import DCCF
    c = E2E
    CAP.enabled = False
    rec = {k: np.array(v) for k, v in c.items() if k != 'seeds'}
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    done, vals = [], []
    try:
        os.makedirs(os.path.join(tmp, 'src'))
        synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', c['user_num'], c['item_num'], c['n_draws'],
                            feat_dim=c['feat_dim'], seed=c['data_seed'])
        os.chdir(os.path.join(tmp, 'src'))
        logging.basicConfig(level=logging.WARNING)
        for seed in c['seeds']:
            torch.manual_seed(seed)
            np.random.seed(seed)
            dl = DataLoader(path='../dataset/', dataset='toy', label='label', sep=',')
            dl.feature_info(include_id=DCCF.include_id, include_item_features=DCCF.include_item_features,
                            include_user_features=DCCF.include_user_features)
            model = DCCF(path=dl.path, dataset=dl.dataset, sentence_model='paraphrase-distilroberta-base-v1', sample_num=10,
                         attribute_num=2, std=0.1, label_min=dl.label_min, label_max=dl.label_max, feature_num=0,
                         user_num=dl.user_num, item_num=dl.item_num, u_vector_size=c['D'], i_vector_size=c['D'], n_layers=1,
                         random_seed=seed, model_path=os.path.join(tmp, 'm.pt'))
            model.apply(model.init_paras)
            dl.drop_neg()
            dp = DataProcessor(dl, model, rank=1, test_neg_n=c['test_neg_n'])
            runner = BaseRunner(optimizer='Adam', learning_rate=c['lr'], epoch=0, batch_size=c['batch_size'], eval_batch_size=128 * 128,
                                dropout=0.2, l2=1e-4, metrics='ndcg@5,recall@5,precision@5', check_epoch=0, early_stop=1)
            v = runner.evaluate(model, dp.get_validation_data(), dp)
            print('seed', seed, 'untrained validation', v, flush=True)
            done.append(seed)
            vals.append([float(x) for x in v])
            rec['seeds'] = np.array(done)
            rec['init_valid'] = np.array(vals)
            np.savez_compressed(os.path.join(outdir, os.environ.get('E2E_OUT', 'e2e_init.npz')), **rec)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)


def gen_fit_rate(outdir):
    """the UNMODIFIED reference's own `BaseRunner.fit` rate (pairs/s = len(train) / seconds of one fit
    epoch, per-epoch negative sampling and batching included, second epoch onward) at D = 64 on the 5k x 5k set, the README
    hyper-parameters, all the container's threads.  Prints one JSON line (recorded in BASELINE.md / profiles/)."""
    import json
    import logging
    import time
    from dccf_amd import synth
    from data_loaders.DataLoader import DataLoader
    from data_processor.DataProcessor import DataProcessor
    from runners.BaseRunner import BaseRunner
    from models.DCCF import DCCF
    c = dict(E2E_CONFIGS['c1'], D=int(os.environ.get('FIT_D', '64')))
    CAP.enabled = False
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    try:
        os.makedirs(os.path.join(tmp, 'src'))
        synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', c['user_num'], c['item_num'], c['n_draws'],
                            feat_dim=c['feat_dim'], seed=c['data_seed'])
        os.chdir(os.path.join(tmp, 'src'))
        logging.basicConfig(level=logging.WARNING)
        torch.manual_seed(2019)
        np.random.seed(2019)
        dl = DataLoader(path='../dataset/', dataset='toy', label='label', sep=',')
        dl.feature_info(include_id=DCCF.include_id, include_item_features=DCCF.include_item_features,
                        include_user_features=DCCF.include_user_features)
        model = DCCF(path=dl.path, dataset=dl.dataset, sentence_model='paraphrase-distilroberta-base-v1', sample_num=10,
                     attribute_num=2, std=0.1, label_min=dl.label_min, label_max=dl.label_max, feature_num=0,
                     user_num=dl.user_num, item_num=dl.item_num, u_vector_size=c['D'], i_vector_size=c['D'], n_layers=1,
                     random_seed=2019, model_path=os.path.join(tmp, 'm.pt'))
        model.apply(model.init_paras)
        dl.drop_neg()
        dp = DataProcessor(dl, model, rank=1, test_neg_n=c['test_neg_n'])
        runner = BaseRunner(optimizer='Adam', learning_rate=0.001, epoch=2, batch_size=128, eval_batch_size=128 * 128,
                            dropout=0.2, l2=1e-4, metrics='ndcg@5,recall@5,precision@5', check_epoch=0, early_stop=1)
        n_epochs = int(os.environ.get('FIT_EPOCHS', '2'))
        secs = []
        for e in range(n_epochs):
            train = dp.get_train_data(epoch=e)
            t0 = time.time()
            runner.fit(model, train, dp, epoch=e)
            secs.append(time.time() - t0)
        n = len(dl.train_df)
        print(json.dumps({'what': 'unmodified reference BaseRunner.fit (src/runners/BaseRunner.py:159-191), CPU', 'D': c['D'],
                          'users': c['user_num'], 'items': c['item_num'], 'train_pairs': n, 'threads': torch.get_num_threads(),
                          'fit_seconds_per_epoch': [round(x, 2) for x in secs], 'pairs_per_s': [round(n / x, 1) for x in secs]}), flush=True)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)


if __name__ == '__main__':
    install_shims()
    which = sys.argv[1:] or ['dccf', 'mf', 'opt', 'batches', 'metrics']
    torch.manual_seed(2019)
    np.random.seed(2019)
    if 'dccf' in which:
        gen_dccf(HERE)
    if 'mf' in which:
        gen_mf(HERE)
    if 'opt' in which:
        gen_opt(HERE)
    if 'batches' in which:
        gen_batches(HERE)
    if 'metrics' in which:
        gen_metrics(HERE)
    if 'dccf_ext' in which:
        gen_dccf_ext(HERE)
    if 'dccf_wide' in which:
        gen_dccf_wide(HERE)
    if 'e2e' in which:
        gen_e2e(HERE)
    if 'e2e_init' in which:
        gen_e2e_init(HERE)
    if 'fit_rate' in which:
        gen_fit_rate(HERE)
