# coding=utf-8
"""Model classes with the reference's contract (src/models/BaseModel.py:16-23), backed by the HIP path.

Same class names, flags, constructor arguments, ``predict`` / ``forward`` return dicts, ``l2``, ``count_variables``,
``save_model`` / ``load_model`` and ``state_dict`` keys and shapes as the reference (so ``.pt`` files interchange:
``uid_embeddings.weight``, ``iid_embeddings.weight``, ``mlp.0.weight``, ``mlp.0.bias``, ``user_bias.weight`` [U,1],
``item_bias.weight`` [I,1], ``global_bias`` []).  What differs is underneath: parameters are views of ONE flat fp32
buffer in HBM (so the dense regularised optimizer step is a single streaming kernel), and ``forward`` in training mode
runs the fused forward + loss + backward kernels and leaves the loss-term gradients in the flat gradient buffer —
there is no autograd graph.  No CPU fallback: without the HIP library or a GPU these classes raise.
"""
import logging
import os
from collections import OrderedDict

import numpy as np
import torch

from dccf_amd import _lib, utils
from dccf_amd.rank_metrics import evaluate_method as _evaluate_method


class _ParamModule(object):
    """What ``model.apply(fn)`` visits: stands in for the reference's nn.Embedding / nn.Linear modules."""

    def __init__(self, kind, weight, bias=None):
        self.kind, self.weight, self.bias = kind, weight, bias


class FusedOptimizer(object):
    """torch.optim.{SGD,Adagrad,Adam}(lr, weight_decay=l2) + the explicit l2 term + clip_grad_value_(50) of
    src/runners/BaseRunner.py:83-107,181-187 as ONE dense kernel over the flat parameter buffer."""

    def __init__(self, model, name, lr, l2, clip=50.0):
        self.model, self.name, self.lr, self.l2, self.clip = model, name.lower(), lr, l2, clip
        self.t = 0
        n = model.flat_p.numel()
        dev = model.flat_p.device
        self.s1 = torch.zeros(n, dtype=torch.float32, device=dev) if self.name != 'gd' else None
        self.s2 = torch.zeros(n, dtype=torch.float32, device=dev) if self.name == 'adam' else None

    def zero_grad(self):
        """Grads are zeroed by the fused step itself (src/runners/BaseRunner.py:178 becomes a no-op)."""
        return

    lazy = None       # _lib.LazyState of the windowed lazy regularisation (DCCF.train_step), when enabled

    def flush(self):
        """Every parameter row up to step t (no-op unless lazy rows are behind): before anything but train_step reads them."""
        if self.lazy is not None:
            self.lazy.flush(self.t)

    def step(self):
        self.flush()
        self.t += 1
        if self.lazy is not None:
            self.lazy.sync_all(self.t)         # the dense pass below brings every row to t
        m = self.model
        segs = getattr(m, 'row_segments', None)
        if segs:     # g of rows the backward did not touch is zero by construction: not read, not re-zeroed
            _lib.dense_opt_step_rows(self.name, m.flat_p, m.flat_g, self.s1, self.s2, self.lr, self.l2, self.l2, self.clip,
                                     self.t, segs)
        else:
            _lib.dense_opt_step(self.name, m.flat_p, m.flat_g, self.s1, self.s2, self.lr, self.l2, self.l2, self.clip, self.t,
                                zero_grad=True)


class BaseModel(object):
    append_id = False
    include_id = True
    include_user_features = True
    include_item_features = True
    include_context_features = False

    @staticmethod
    def parse_model_args(parser, model_name='BaseModel'):
        parser.add_argument('--model_path', type=str, default='../model/%s/%s.pt' % (model_name, model_name),
                            help='Model save path.')
        return parser

    @staticmethod
    def evaluate_method(p, data, metrics):
        """src/models/BaseModel.py:55-128."""
        return _evaluate_method(p, data, metrics)

    @staticmethod
    def init_paras(m):
        """src/models/BaseModel.py:130-142: N(0, 0.01) for every Embedding weight and Linear weight AND bias."""
        if m.kind in ('linear', 'embedding'):
            torch.nn.init.normal_(m.weight, mean=0.0, std=0.01)
            if m.kind == 'linear' and m.bias is not None:
                torch.nn.init.normal_(m.bias, mean=0.0, std=0.01)

    def __init__(self, label_min, label_max, feature_num, random_seed=2018, model_path='../model/Model/Model.pt'):
        self.label_min, self.label_max, self.feature_num = label_min, label_max, feature_num
        self.random_seed, self.model_path = random_seed, model_path
        torch.manual_seed(self.random_seed)
        self.device = utils.device()
        self.ctx = _lib.Context(self.device.index or 0)
        self.training = False
        self._call = 0            # counter of forwards: the `step` word of the Philox streams
        self._specs = OrderedDict()
        self._modules = []
        self._init_weights()
        self._allocate()
        self.total_parameters = self.count_variables()
        logging.info('# of params: %d' % self.total_parameters)
        self.optimizer = None

    # ---- parameter plumbing
    def _declare(self, name, shape, init='default'):
        self._specs[name] = (tuple(shape), init)

    def _allocate(self):
        sizes = [int(np.prod(s)) if len(s) else 1 for s, _ in self._specs.values()]
        pads = [(n + 255) // 256 * 256 for n in sizes]      # every parameter starts on a 256-float boundary
        self.flat_p = torch.zeros(sum(pads), dtype=torch.float32, device=self.device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.params, self.grads, self.offsets = OrderedDict(), OrderedDict(), OrderedDict()
        o = 0
        for (name, (shape, init)), n, pd in zip(self._specs.items(), sizes, pads):
            self.params[name] = self.flat_p[o:o + n].view(shape)
            self.grads[name] = self.flat_g[o:o + n].view(shape)
            self.offsets[name] = o
            o += pd
            t = self.params[name]
            # torch defaults of the layers the reference builds, overwritten by main.py's model.apply(init_paras)
            if init == 'embedding':
                torch.nn.init.normal_(t)
            elif init == 'linear_w':
                torch.nn.init.kaiming_uniform_(t, a=5 ** 0.5)
            elif init == 'linear_b':
                bound = 1.0 / (self._specs[name.replace('bias', 'weight')][0][1] ** 0.5)
                torch.nn.init.uniform_(t, -bound, bound)
            elif isinstance(init, float):
                t.fill_(init)

    def apply(self, fn):
        for m in self._modules:
            fn(m)
        return self

    def parameters(self):
        return list(self.params.values())

    def views_of(self, flat):
        """The per-parameter views of a flat tensor laid out like flat_p (an optimizer's state buffer), by parameter name."""
        out = OrderedDict()
        for name, v in self.params.items():
            o = self.offsets[name]
            out[name] = flat[o:o + v.numel()].view(v.shape)
        return out

    def named_parameters(self):
        return list(self.params.items())

    def _flush(self):
        """Rows the lazy regularisation of a train_step left behind are brought up to date (no-op otherwise): whatever reads or
        writes the parameters outside train_step calls this first."""
        o = getattr(self, 'optimizer', None)
        if o is not None and getattr(o, 'flush', None):
            o.flush()

    def state_dict(self):
        self._flush()
        return OrderedDict((k, v.detach().clone()) for k, v in self.params.items())

    def load_state_dict(self, sd):
        self._flush()
        missing = [k for k in self.params if k not in sd]
        extra = [k for k in sd if k not in self.params]
        if missing or extra:
            raise RuntimeError('state_dict mismatch: missing %s unexpected %s' % (missing, extra))
        for k, v in self.params.items():
            v.copy_(sd[k].to(self.device).view(v.shape))

    def count_variables(self):
        return sum(p.numel() for p in self.params.values())

    def l2(self):
        """src/models/BaseModel.py:179-187: sum over all parameters of p^2 (pad elements of the flat buffer are zero)."""
        return _lib.sumsq(self.flat_p)[0]

    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    def cuda(self):
        return self

    def __call__(self, feed_dict):
        return self.forward(feed_dict)

    def save_model(self, model_path=None):
        """src/models/BaseModel.py:224-236."""
        model_path = model_path or self.model_path
        sd = self.state_dict()
        if utils.is_rank0():             # (several GPUs: the replicas are bit-identical, rank 0 writes, the others wait for the file)
            d = os.path.dirname(model_path)
            if d and not os.path.exists(d):
                os.makedirs(d)
            torch.save(OrderedDict((k, v.cpu()) for k, v in sd.items()), model_path)
            logging.info('Save model to ' + model_path)
        utils.barrier()

    def load_model(self, model_path=None):
        """src/models/BaseModel.py:238-248."""
        model_path = model_path or self.model_path
        self.load_state_dict(torch.load(model_path, map_location='cpu'))
        self.eval()
        logging.info('Load model from ' + model_path)

    def _next_step(self):
        self._call += 1
        return self._call

    def _init_weights(self):
        raise NotImplementedError

    def predict(self, feed_dict):
        raise NotImplementedError

    def forward(self, feed_dict):
        raise NotImplementedError


class RecModel(BaseModel):
    """src/models/RecModel.py:9-48 — plain MF dot."""
    append_id = True
    include_id = False
    include_user_features = False
    include_item_features = False
    kind = 'RecModel'

    @staticmethod
    def parse_model_args(parser, model_name='RecModel'):
        parser.add_argument('--u_vector_size', type=int, default=64, help='Size of user vectors.')
        parser.add_argument('--i_vector_size', type=int, default=64, help='Size of item vectors.')
        return BaseModel.parse_model_args(parser, model_name)

    def __init__(self, label_min, label_max, feature_num, user_num, item_num, u_vector_size, i_vector_size, random_seed,
                 model_path):
        self.u_vector_size, self.i_vector_size = u_vector_size, i_vector_size
        assert self.u_vector_size == self.i_vector_size
        self.ui_vector_size = self.u_vector_size
        self.user_num, self.item_num = int(user_num), int(item_num)
        BaseModel.__init__(self, label_min=label_min, label_max=label_max, feature_num=feature_num,
                           random_seed=random_seed, model_path=model_path)

    def _init_weights(self):
        self._declare('uid_embeddings.weight', (self.user_num, self.ui_vector_size), 'embedding')
        self._declare('iid_embeddings.weight', (self.item_num, self.ui_vector_size), 'embedding')

    def _allocate(self):
        BaseModel._allocate(self)
        p = self.params
        self._modules = [_ParamModule('embedding', p[k]) for k in p if k.endswith('embeddings.weight') or k.endswith('_bias.weight')]
        # one "touched" byte per embedding row (set by the backward) for the row-aware dense optimizer step; the bias
        # vectors and the global bias stay dense
        self.touchedP = self.touchedQ = None
        if self.ui_vector_size % 4 == 0 and 4 <= self.ui_vector_size <= 256:      # (rows of whole float4 slots; see DCCF._allocate)
            self.touchedP = torch.zeros((self.user_num + 3) // 4 * 4, dtype=torch.uint8, device=self.device)[:self.user_num]
            self.touchedQ = torch.zeros((self.item_num + 3) // 4 * 4, dtype=torch.uint8, device=self.device)[:self.item_num]
            self.row_segments = [(self.offsets['uid_embeddings.weight'], self.user_num, self.ui_vector_size, self.touchedP),
                                 (self.offsets['iid_embeddings.weight'], self.item_num, self.ui_vector_size, self.touchedQ)]

    propensity = None
    M = 0.1

    def _struct(self):
        p = self.params
        if self.kind == 'RecModel':
            return _lib.mf_struct(self.kind, p['uid_embeddings.weight'], p['iid_embeddings.weight'])
        return _lib.mf_struct(self.kind, p['uid_embeddings.weight'], p['iid_embeddings.weight'],
                              p['user_bias.weight'].view(-1), p['item_bias.weight'].view(-1), p['global_bias'].view(-1),
                              self.propensity, self.M)

    def predict(self, feed_dict):
        self._flush()
        pred = _lib.mf_predict(self._struct(), feed_dict['X'].contiguous())
        return {'prediction': pred, 'check': []}

    def eval(self):
        self._flush()
        return BaseModel.eval(self)

    def parameters(self):
        self._flush()
        return BaseModel.parameters(self)

    def named_parameters(self):
        self._flush()
        return BaseModel.named_parameters(self)

    def l2(self):
        self._flush()
        return BaseModel.l2(self)

    def forward(self, feed_dict):
        """BaseModel.forward (src/models/BaseModel.py:203-219).  In training mode the backward of the loss term is
        fused in and its gradients are accumulated into ``flat_g``."""
        if not self.training:
            out = self.predict(feed_dict)
            return out
        self._flush()
        g = self.grads
        X = feed_dict['X'].contiguous()
        if self.kind == 'RecModel':
            pred, loss = _lib.mf_train_fwdbwd(self.ctx, self._struct(), X, feed_dict['Y'], feed_dict['rank'],
                                              g['uid_embeddings.weight'], g['iid_embeddings.weight'],
                                              touchedP=self.touchedP, touchedQ=self.touchedQ)
        else:
            pred, loss = _lib.mf_train_fwdbwd(self.ctx, self._struct(), X, feed_dict['Y'], feed_dict['rank'],
                                              g['uid_embeddings.weight'], g['iid_embeddings.weight'],
                                              g['user_bias.weight'].view(-1), g['item_bias.weight'].view(-1),
                                              g['global_bias'].view(-1), touchedP=self.touchedP, touchedQ=self.touchedQ)
        return {'prediction': pred, 'check': [], 'loss': loss[0]}

    lazy_K = int(os.environ.get('DCCF_LAZY_K', '8'))          # 0: forward + the dense optimizer pass of every step

    def train_step(self, feed_dict, overlap=0, X_next=None):
        """zero_grad + forward + backward + `+ l2` + clip + optimizer.step() of src/runners/BaseRunner.py:172-188 for the MF family
        as ONE library call (mf_train_step) under the windowed lazy regularisation: the dense pass over all
        (user_num + item_num) x D parameters — 80 % of an IPSBiasedMF step at batch 128 — shrinks to one lazy_K-th per step;
        results equal the dense step (untouched rows bit for bit).  Without row segments (embedding sizes other than 16 / 32 / 64
        / 128), with lazy_K < 2 or above 8192 batch rows: the two calls of the plain step."""
        o = self.optimizer
        X = feed_dict['X'].contiguous()
        N = X.shape[0]
        if self.lazy_K < 2 or not getattr(self, 'row_segments', None) or N > 8192 or N < 1:
            if o.lazy is not None:
                o.flush()
                o.lazy = None
                self._mf_opt_key = None
            out = self.forward(feed_dict)
            o.step()
            return out
        key = (o.name, o.lr, o.l2, o.clip)
        if getattr(self, '_mf_opt_key', None) != key or o.lazy is None:
            o.flush()
            self._mf_opt = _lib.opt_struct(o.name, self.flat_p, self.flat_g, o.s1, o.s2, o.lr, o.l2, o.l2, o.clip, self.row_segments, 0)
            o.lazy = _lib.LazyState(self._mf_opt, self.lazy_K, self.user_num + self.item_num, 2 * 8192, o.lr, self.device)
            o.lazy.sync_all(o.t)
            self._mf_ids = torch.empty(2 * 8192, dtype=torch.int32, device=self.device)
            self._mf_loss = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._mf_opt_key = key
        t1 = o.t + 1                 # (the counter moves only once the library took the step: a raise leaves no phantom step)
        o.lazy.cover(t1)
        o.lazy.dirty = True
        g = self.grads
        bias = self.kind != 'RecModel'
        pred, loss = _lib.mf_train_step(self.ctx, self._struct(), X, feed_dict['Y'], feed_dict['rank'],
                                        g['uid_embeddings.weight'], g['iid_embeddings.weight'],
                                        g['user_bias.weight'].view(-1) if bias else None,
                                        g['item_bias.weight'].view(-1) if bias else None,
                                        g['global_bias'].view(-1) if bias else None, self._mf_opt, t1, self._mf_ids,
                                        loss=self._mf_loss)
        o.t = t1
        return {'prediction': pred, 'check': [], 'loss': loss[0]}

    def full_matrix(self):
        """README.md:28-30: the full predicted user x item matrix (the exposure probabilities DCCF loads)."""
        self._flush()
        return _lib.mf_predict_full(self._struct(), device=self.device)


class BiasedMF(RecModel):
    """src/models/BiasedMF.py:9-33."""
    kind = 'BiasedMF'

    def _init_weights(self):
        RecModel._init_weights(self)
        self._declare('user_bias.weight', (self.user_num, 1), 'embedding')
        self._declare('item_bias.weight', (self.item_num, 1), 'embedding')
        self._declare('global_bias', (), 0.1)


class IPSBiasedMF(BiasedMF):
    """src/models/IPSBiasedMF.py:12-57."""
    kind = 'IPSBiasedMF'

    @staticmethod
    def parse_model_args(parser, model_name='IPSBiasedMF'):
        parser.add_argument('--M', type=float, default=0.1, help='minimum propensity to avoid high variance.')
        return RecModel.parse_model_args(parser, model_name)

    def __init__(self, path, dataset, M, label_min, label_max, feature_num, user_num, item_num, u_vector_size,
                 i_vector_size, random_seed, model_path):
        self.path, self.dataset = path, dataset
        RecModel.__init__(self, label_min=label_min, label_max=label_max, feature_num=feature_num, user_num=user_num,
                          item_num=item_num, u_vector_size=u_vector_size, i_vector_size=i_vector_size,
                          random_seed=random_seed, model_path=model_path)
        self.M = M
        self.propensity = utils.numpy_to_torch(np.load(os.path.join(path, dataset + utils.PROPENSITY_SUFFIX)).astype(np.float32))


class DMF(RecModel):
    """Only the flag and constructor plumbing DCCF inherits (src/models/DMF.py:11-23); DMF itself is not constructible
    from the reference's main.py and is outside the hot path."""

    @staticmethod
    def parse_model_args(parser, model_name='DMF'):
        parser.add_argument('--n_layers', type=int, default=1, help='Number of mlp layers.')
        return RecModel.parse_model_args(parser, model_name)


class DCCF(DMF):
    """src/models/DCCF.py:14-127."""
    kind = 'DCCF'

    @staticmethod
    def parse_model_args(parser, model_name='DCCF'):
        parser.add_argument('--sentence-model', type=str, default='paraphrase-distilroberta-base-v1',
                            help='the name of sentence model')
        parser.add_argument('--sample-num', type=int, default=10, help='the number of sampled items')
        parser.add_argument('--attribute-num', type=int, default=2, help='the number of item features')
        parser.add_argument('--std', type=float, default=0.1, help='std of feature distribution')
        return DMF.parse_model_args(parser, model_name)

    def __init__(self, path, dataset, sentence_model, sample_num, attribute_num, std, label_min, label_max, feature_num,
                 user_num, item_num, u_vector_size, i_vector_size, n_layers, random_seed, model_path,
                 feature_embedding=None, expo_prob=None, ips_factors=None):
        """``feature_embedding`` / ``expo_prob`` tensors may be passed instead of the .npy files (benchmarks);
        ``ips_factors`` (dict P,Q,bu,bi,prop,b0,M) replaces the dense exposure matrix by on-the-fly IPSBiasedMF scores."""
        self.path, self.dataset, self.sentence_model = path, dataset, sentence_model
        self.sample_num, self.attribute_num, self.std = sample_num, attribute_num, std
        self.n_layers = int(n_layers)
        if not 1 <= self.n_layers <= 8:
            raise ValueError('--n_layers must be in [1, 8] (src/models/DMF.py:14; the HIP path holds up to 7 extra D x D layers)')
        if u_vector_size > 256 or (u_vector_size > 128 and self.n_layers > 1):
            raise ValueError('embedding sizes up to 256 (src/models/RecModel.py:17-27 accepts any; above 128 with --n_layers 1 only)')
        self._feat_in, self._expo_in, self.ips_factors = feature_embedding, expo_prob, ips_factors
        RecModel.__init__(self, label_min=label_min, label_max=label_max, feature_num=feature_num, user_num=user_num,
                          item_num=item_num, u_vector_size=u_vector_size, i_vector_size=i_vector_size,
                          random_seed=random_seed, model_path=model_path)

    def _init_weights(self):
        """src/models/DCCF.py:47-64."""
        if self._feat_in is not None:
            self.feature_embedding = self._feat_in.to(self.device, torch.float32).contiguous()
        else:
            f = np.load(os.path.join(self.path, self.dataset + '_' + self.sentence_model + '.npy'))
            self.feature_embedding = utils.numpy_to_torch(f.astype(np.float32))
        if self._expo_in is not None:
            self.expo_prob = self._expo_in.to(self.device, torch.float32).contiguous()
        elif self.ips_factors is not None:
            self.expo_prob = None
        else:
            e = np.load(os.path.join(self.path, self.dataset + utils.EXPO_SUFFIX), mmap_mode='r')
            self.expo_prob = torch.empty(e.shape, dtype=torch.float32, device=self.device)
            rows = max(1, (256 << 20) // (4 * e.shape[1]))          # stream the U x I matrix to HBM in 256 MiB slabs
            for r0 in range(0, e.shape[0], rows):
                self.expo_prob[r0:r0 + rows].copy_(torch.from_numpy(np.array(e[r0:r0 + rows], dtype=np.float32)))
        D, F = self.ui_vector_size, self.feature_embedding.shape[1]
        self._declare('uid_embeddings.weight', (self.user_num, D), 'embedding')
        self._declare('iid_embeddings.weight', (self.item_num, D), 'embedding')
        self._declare('mlp.0.weight', (D, D + F), 'linear_w')
        self._declare('mlp.0.bias', (D,), 'linear_b')
        for k in range(1, self.n_layers):        # src/models/DCCF.py:61-62: n_layers - 1 more Linear(D, D)
            self._declare('mlp.%d.weight' % k, (D, D), 'linear_w')
            self._declare('mlp.%d.bias' % k, (D,), 'linear_b')

    def _extra(self, views):
        return [(views['mlp.%d.weight' % k], views['mlp.%d.bias' % k]) for k in range(1, self.n_layers)]

    def _allocate(self):
        BaseModel._allocate(self)
        p = self.params
        self._modules = [_ParamModule('embedding', p['uid_embeddings.weight']), _ParamModule('embedding', p['iid_embeddings.weight'])] + \
                        [_ParamModule('linear', p['mlp.%d.weight' % k], p['mlp.%d.bias' % k]) for k in range(self.n_layers)]
        # one "touched" byte per embedding row, set by the backward, consumed by the row-aware dense optimizer step
        # (padded to whole 32-bit words: the overlapped step marks them with word atomics)
        self.touchedU = torch.zeros((self.user_num + 3) // 4 * 4, dtype=torch.uint8, device=self.device)[:self.user_num]
        self.touchedV = torch.zeros((self.item_num + 3) // 4 * 4, dtype=torch.uint8, device=self.device)[:self.item_num]
        D = self.ui_vector_size
        if D % 4 == 0 and 4 <= D <= 256:
            # 16 / 32 / 64 / 128: every form of the optimizer pass.  Other multiples of 4 (src/models/RecModel.py:17-27 accepts any
            # size): the windowed lazy regularisation of train_step works on rows of any such width; dense calls on these segments
            # run the plain dense pass
            self.row_segments = [(self.offsets['uid_embeddings.weight'], self.user_num, D, self.touchedU),
                                 (self.offsets['iid_embeddings.weight'], self.item_num, D, self.touchedV)]
        else:        # rows that are not whole float4 slots (D = 7 ...): no row segments, the plain dense step throughout
            self.row_segments = []
            self.touchedU = self.touchedV = None

    def _struct(self):
        """The C view of the model; parameter storage never moves, so it is built once."""
        if getattr(self, '_ms', None) is None:
            p = self.params
            self._ms = _lib.model_struct(p['uid_embeddings.weight'], p['iid_embeddings.weight'], p['mlp.0.weight'],
                                         p['mlp.0.bias'], self.feature_embedding, self.expo_prob, self.sample_num,
                                         self.attribute_num, self.std, ips=self.ips_factors, extra=self._extra(p))
            self._rs = _lib.rand_struct(seed=self.random_seed, step=0)
            self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        return self._ms

    def _rand(self, feed_dict):
        inj = feed_dict.get('inject')
        if inj is not None:   # parity tests: the reference's captured draws
            return _lib.rand_struct(sample_item=inj['sample_item'], noise=inj['noise'], keep=inj.get('keep'))
        if os.environ.get('DCCF_TORCH_DRAWS') == '1':
            # A/B diagnostic (scripts/e2e_ab.py, arm torch_draws): the draws of DCCF.predict (src/models/DCCF.py:72,87,94) come from
            # torch's generator through the INJECTED kernel path instead of the Philox streams — same distributions, another RNG
            X = feed_dict['X']
            N, S, A, D = X.shape[0], self.sample_num, self.attribute_num, self.ui_vector_size
            Ld, F, p = N * (S + 1) * A, self.feature_embedding.shape[1], float(feed_dict['dropout'])
            keep = None
            if p > 0:
                keep = (torch.rand((self.n_layers, Ld, D), device=X.device) >= p).to(torch.uint8)
            self._call += 1
            return _lib.rand_struct(sample_item=torch.randint(self.item_num, (N, S), device=X.device),
                                    noise=torch.randn((Ld, F), device=X.device) * self.std, keep=keep)
        self._struct()
        self._rs.step = self._next_step()
        return self._rs

    eval_noise = 'full'       # 'projected': evaluation draws the D-dim projected noise (same distribution, K = D instead of F)

    def begin_eval(self):
        """Called once per evaluation pass: with eval_noise == 'projected' refreshes the tables of dccf_predict_projected
        (the parameters changed since the last pass)."""
        if self.eval_noise == 'projected':
            self._proj = _lib.dccf_eval_prepare(self.ctx, self._struct(), *(getattr(self, '_proj', None) or (None, None)))

    def predict(self, feed_dict):
        """src/models/DCCF.py:66-107.  Fresh candidates and noise on every call, also in eval mode, as in the reference."""
        self._flush()
        if (self.eval_noise == 'projected' and not self.training and feed_dict.get('inject') is None
                and getattr(self, '_proj', None) is not None):
            pred = _lib.dccf_predict_projected(self.ctx, self._struct(), self._rand(feed_dict), feed_dict['X'].contiguous(),
                                               feed_dict['dropout'], *self._proj)
            return {'prediction': pred, 'check': [('prediction', pred)]}
        pred = _lib.dccf_predict(self.ctx, self._struct(), self._rand(feed_dict), feed_dict['X'].contiguous(),
                                 feed_dict['dropout'])
        return {'prediction': pred, 'check': [('prediction', pred)]}

    def forward(self, feed_dict):
        """src/models/DCCF.py:109-127 (+ the backward of the loss term when training)."""
        if not self.training:
            return self.predict(feed_dict)
        self._flush()
        g = self.grads
        ms = self._struct()
        pred, loss = _lib.dccf_train_fwdbwd(self.ctx, ms, self._rand(feed_dict), feed_dict['X'].contiguous(),
                                            feed_dict['Y'], feed_dict['rank'], feed_dict['dropout'],
                                            g['uid_embeddings.weight'], g['iid_embeddings.weight'], g['mlp.0.weight'],
                                            g['mlp.0.bias'], loss=self._loss, touchedU=self.touchedU, touchedV=self.touchedV,
                                            gextra=self._extra(g))
        return {'prediction': pred, 'check': [('prediction', pred)], 'loss': loss[0]}

    lazy_K = int(os.environ.get('DCCF_LAZY_K', '8'))      # window count of the lazy regularisation; 0 = dense pass every step

    def _flush(self):
        if self.optimizer is not None and getattr(self.optimizer, 'flush', None):
            self.optimizer.flush()

    def eval(self):
        self._flush()             # evaluation reads every row
        return BaseModel.eval(self)

    def state_dict(self):
        self._flush()
        return BaseModel.state_dict(self)

    def parameters(self):
        self._flush()
        return BaseModel.parameters(self)

    def named_parameters(self):
        self._flush()
        return BaseModel.named_parameters(self)

    def l2(self):
        self._flush()
        return BaseModel.l2(self)

    def train_step(self, feed_dict, overlap=0, X_next=None):
        """zero_grad + forward + backward + `+ l2` + clip + optimizer.step() of src/runners/BaseRunner.py:172-188 as ONE
        library call (dccf_train_step).  ``overlap``: 0 = optimizer pass after the backward; 1 = the pass over the
        embedding rows this batch does not touch runs on a side stream beside forward/backward.  Same arithmetic per
        element either way.  ``X_next``: the batch the NEXT train_step will be given (the same tensor, not a copy) — its
        candidates are then drawn inside this step's optimizer launch; results are identical with or without it."""
        o = self.optimizer
        key = (o.name, o.lr, o.l2, o.clip, int(overlap))
        if getattr(self, '_opt_struct_key', None) != key:
            o.flush()
            self._opt_struct = _lib.opt_struct(o.name, self.flat_p, self.flat_g, o.s1, o.s2, o.lr, o.l2, o.l2, o.clip,
                                               self.row_segments, overlap)
            self._opt_struct_key = key
            o.lazy = None
            # (at 2B > 2048 the step touches tens of thousands of rows and the dense pass is a small part of it: measured 5.05 M
            # pairs/s lazy against 5.3 M dense at B = 4096 — the dense pass stays there)
            # (injected draws — the parity tests' golden cases — take the same lazy path as the fused ones)
            if (self.lazy_K >= 2 and not overlap and self.row_segments
                    and feed_dict['X'].shape[0] <= int(os.environ.get('DCCF_LAZY_MAX_N', '2048'))):
                # windowed lazy regularisation: only 1 / lazy_K of the untouched rows is streamed per step (dccf_opt_t.lazy_*)
                rows = self.user_num + self.item_num
                o.lazy = _lib.LazyState(self._opt_struct, self.lazy_K, rows, 16 + 4 * 16384 * (self.sample_num + 2), o.lr, self.device)
                o.lazy.sync_all(o.t)
        g = self.grads
        t1 = o.t + 1                     # (the counter moves only once the library took the step: a raise leaves no phantom step)
        if o.lazy is not None:
            if feed_dict['X'].shape[0] * (self.sample_num + 2) > o.lazy.list_cap:
                raise RuntimeError('batch too large for the lazy optimizer row list')
            o.lazy.cover(t1)
            o.lazy.dirty = True
        rs = self._rand(feed_dict)
        if X_next is not None and (overlap or feed_dict.get('inject') is not None or not X_next.is_contiguous()):
            X_next = None
        pred, loss = _lib.dccf_train_step(self.ctx, self._struct(), rs, feed_dict['X'].contiguous(),
                                          feed_dict['Y'], feed_dict['rank'], feed_dict['dropout'],
                                          g['uid_embeddings.weight'], g['iid_embeddings.weight'], g['mlp.0.weight'],
                                          g['mlp.0.bias'], self._opt_struct, t1, loss=self._loss, touchedU=self.touchedU,
                                          touchedV=self.touchedV, X_next=X_next, step_next=self._call + 1, gextra=self._extra(g))
        o.t = t1
        return {'prediction': pred, 'check': [('prediction', pred)], 'loss': loss[0]}


class StepGraph(object):
    """One DCCF training step — forward + loss + backward kernels, the row-aware dense optimizer step, and the advance of
    a device-side step counter — captured ONCE as a hipGraph and replayed for every full batch of a run.

    What varies from step to step lives on the device: with k = *k_dev the kernels read batch ``full[k % nb]``, draw
    from the Philox streams at counter ``step0 + k`` and the optimizer uses ``t = t0 + k`` (bias corrections computed in
    the kernel), so a replay needs no parameter update and costs one host call.  The partial last batch of an epoch runs
    eagerly through the same counter."""

    def __init__(self, model, opt, nb, rows, dropout):
        self.model, self.opt, self.nb, self.rows, self.dropout = model, opt, nb, rows, dropout
        opt.flush()                      # (rows a lazy train_step left behind; the graph replays the dense pass)
        opt.lazy, model._opt_struct_key = None, None
        dev = model.device
        self.k = torch.zeros(1, dtype=torch.int64, device=dev)
        self.k_host = 0
        self.full = torch.zeros((nb, rows, 2), dtype=torch.int64, device=dev)
        B = rows // 2
        self.Y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
        self.pred = torch.empty(rows, dtype=torch.float32, device=dev)
        self.ms = model._struct()
        self.step0, self.t0 = model._call + 1, opt.t + 1
        self.rs = _lib.rand_struct(seed=model.random_seed, step=self.step0, k_dev=self.k, x_stride=rows * 2, x_steps=nb)
        model.ctx.reserve(rows, model.ui_vector_size, model.feature_embedding.shape[1], model.sample_num, model.attribute_num)
        self.graph = None

    def _body(self, X, Y, rs, pred):
        m, o, g = self.model, self.opt, self.model.grads
        _lib.dccf_train_fwdbwd(m.ctx, self.ms, rs, X, Y, 1, self.dropout, g['uid_embeddings.weight'], g['iid_embeddings.weight'],
                               g['mlp.0.weight'], g['mlp.0.bias'], pred=pred, loss=m._loss, touchedU=m.touchedU,
                               touchedV=m.touchedV, gextra=m._extra(g))
        _lib.dense_opt_step_rows(o.name, m.flat_p, m.flat_g, o.s1, o.s2, o.lr, o.l2, o.l2, o.clip, self.t0, m.row_segments,
                                 k_dev=self.k)
        _lib.advance(self.k)

    def load_epoch(self, full):
        """Copies the epoch's [nb, 2B, 2] batches so that the j-th batch of the epoch sits where step k_host + j looks."""
        assert tuple(full.shape) == tuple(self.full.shape)
        slot = (self.k_host + torch.arange(self.nb, device=full.device)) % self.nb
        self.full[slot] = full

    def _count(self):
        self.k_host += 1
        self.model._call += 1
        self.opt.t += 1

    def step(self):
        """The next full batch (graph replay; the very first call runs eagerly as the capture's warm-up)."""
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._body(self.full[0], self.Y, self.rs, self.pred)          # a real step (k = k_host), on a side stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body(self.full[0], self.Y, self.rs, self.pred)
        else:
            self.graph.replay()
        self._count()

    def tail(self, X):
        """A partial batch [2r, 2], eagerly, through the same device counter (so streams and Adam's t stay in step)."""
        r = X.shape[0] // 2
        rs = _lib.rand_struct(seed=self.model.random_seed, step=self.step0, k_dev=self.k, x_stride=0, x_steps=1)
        Y = torch.cat([self.Y[:r], self.Y[self.rows // 2:self.rows // 2 + r]])
        pred = torch.empty(2 * r, dtype=torch.float32, device=X.device)
        self._body(X.contiguous(), Y, rs, pred)
        self._count()
        return pred

    def out_dict(self):
        return {'prediction': self.pred, 'check': [('prediction', self.pred)], 'loss': self.model._loss[0]}
