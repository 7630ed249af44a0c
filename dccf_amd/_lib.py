# coding=utf-8
"""ctypes binding of libdccf_hip.so (include/dccf_hip.h).  No torch types cross the C ABI: wrappers pass
``tensor.data_ptr()``, sizes and the current HIP stream.  There is no CPU fallback: if the library is missing or a
call fails, a RuntimeError is raised."""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libdccf_hip.so')

EXPORTS = ['dccf_ctx_create', 'dccf_ctx_destroy', 'dccf_ctx_set_deterministic', 'dccf_ctx_reserve', 'dccf_last_error', 'dccf_abi_version',
           'dccf_predict', 'dccf_train_fwdbwd', 'dccf_dense_opt_step', 'dccf_sumsq', 'mf_predict', 'mf_train_fwdbwd', 'mf_train_step',
           'mf_predict_full', 'dccf_sample_train_negatives', 'dccf_debug_candidates', 'dccf_debug_noise',
           'dccf_debug_keep', 'dccf_debug_keep_layer', 'dccf_debug_opt_elem', 'dccf_debug_workspace', 'dccf_profile', 'dccf_profile_read',
           'shard_pack_rows', 'shard_unpack_rows', 'shard_scatter_add', 'dccf_dense_opt_step_rows', 'dccf_dense_opt_step_dev', 'dccf_advance', 'rank_eval_topk', 'dccf_train_step', 'dp_buffer_words',
           'dp_export_touched', 'dp_import_touched', 'dp_mark_global', 'dccf_dense_opt_phase', 'dccf_ctx_side_stream', 'dp_import_apply',
           'dccf_sample_eval_negatives', 'dccf_eval_prepare', 'dccf_predict_projected', 'dccf_dp_local', 'dccf_dp_overlap',
           'dccf_dp_finish', 'dccf_ctx_prepared_steps', 'dccf_ctx_hosted_rows', 'shard_pack_multi', 'shard_unpack_multi', 'dccf_build_epoch_batches', 'dccf_lazy_scalars', 'dccf_lazy_flush', 'dccf_lazy_catchup_rows', 'dccf_lazy_opt_step', 'dccf_comm_unique_id',
           'dccf_comm_create', 'dccf_comm_destroy', 'dccf_comm_all_to_all_rows', 'dccf_comm_all_to_all_rows2', 'dccf_comm_all_reduce_sum']

ABI_VERSION = 6
OPT_KIND = {'gd': 0, 'adagrad': 1, 'adam': 2}
MF_KIND = {'RecModel': 0, 'BiasedMF': 1, 'IPSBiasedMF': 2}

_f = C.c_void_p   # device pointers travel as plain addresses


class ModelT(C.Structure):
    _fields_ = [('user_num', C.c_int64), ('item_num', C.c_int64), ('D', C.c_int32), ('F', C.c_int32),
                ('S', C.c_int32), ('A', C.c_int32), ('std', C.c_float), ('reserved', C.c_float),
                ('U', _f), ('V', _f), ('W', _f), ('b', _f), ('feat', _f), ('expo', _f),
                ('ipsP', _f), ('ipsQ', _f), ('ipsBu', _f), ('ipsBi', _f), ('ipsProp', _f),
                ('ipsB0', C.c_float), ('ipsM', C.c_float), ('ipsD', C.c_int32), ('n_extra', C.c_int32),
                ('Wl', _f * 7), ('bl', _f * 7), ('expo_gathered', _f)]


class RandT(C.Structure):
    _fields_ = [('mode', C.c_int32), ('reserved', C.c_int32), ('sample_item', _f), ('noise', _f), ('keep', _f),
                ('seed', C.c_uint64), ('step', C.c_uint64), ('k_dev', _f), ('x_stride', C.c_int64), ('x_steps', C.c_int64)]


class GradsT(C.Structure):
    _fields_ = [('gU', _f), ('gV', _f), ('gW', _f), ('gb', _f), ('touchedU', _f), ('touchedV', _f),
                ('gWl', _f * 7), ('gbl', _f * 7)]


class OptT(C.Structure):
    _fields_ = [('kind', C.c_int32), ('overlap', C.c_int32), ('p', _f), ('g', _f), ('s1', _f), ('s2', _f), ('n', C.c_int64),
                ('lr', C.c_float), ('wd', C.c_float), ('l2', C.c_float), ('clip', C.c_float), ('step', C.c_int64),
                ('nseg', C.c_int32), ('reserved', C.c_int32), ('seg_begin', _f), ('seg_rows', _f), ('seg_width', _f),
                ('seg_flags', _f),
                # windowed lazy regularisation (include/dccf_hip.h: dccf_opt_t.lazy_*)
                ('lazy_K', C.c_int32), ('lazy_nscal', C.c_int32), ('lazy_last', _f), ('lazy_claim', _f), ('lazy_list', _f),
                ('lazy_cnt', _f), ('lazy_scal', _f), ('lazy_t0', C.c_int64), ('lazy_list_cap', C.c_int64), ('lazy_id', C.c_int64), ('lazy_host', _f)]


class DpT(C.Structure):
    _fields_ = [('G', C.c_int32), ('rank', C.c_int32), ('D', C.c_int32), ('S', C.c_int32), ('cap', C.c_int64),
                ('dense_begin', C.c_int64), ('item_num', C.c_int64), ('seed', C.c_uint64), ('buf', _f), ('bufs', _f),
                ('loss', _f), ('loss_sum', _f), ('gflagsU', _f), ('gflagsV', _f), ('gflagsU2', _f), ('gflagsV2', _f),
                ('lflagsU', _f), ('lflagsV', _f), ('llist', _f), ('lcnt', _f), ('ctx', _f), ('pmask', _f), ('pwhere', _f), ('segU', C.c_int32), ('segV', C.c_int32),
                ('glist', _f), ('gcnt', _f), ('mask', _f), ('where', _f)]


class DpNextT(C.Structure):
    _fields_ = [('ctx', _f), ('model', C.POINTER(ModelT)), ('X_next', _f), ('X_all_next', _f), ('N', C.c_int64),
                ('step0_next', C.c_uint64)]


class ShardJobT(C.Structure):
    _fields_ = [('idx', _f), ('dst', _f), ('n', C.c_int64), ('tables', _f * 4), ('widths', C.c_int32 * 4),
                ('ntables', C.c_int32), ('ld', C.c_int32), ('buf', _f)]


class MFModelT(C.Structure):
    _fields_ = [('user_num', C.c_int64), ('item_num', C.c_int64), ('D', C.c_int32), ('kind', C.c_int32),
                ('P', _f), ('Q', _f), ('bu', _f), ('bi', _f), ('b0', _f), ('prop', _f), ('M', C.c_float),
                ('reserved', C.c_float)]


class MFGradsT(C.Structure):
    _fields_ = [('gP', _f), ('gQ', _f), ('gbu', _f), ('gbi', _f), ('gb0', _f), ('touchedP', _f), ('touchedQ', _f)]


_lib = None


def load():
    """Loads the shared library (built in-tree by dccf_amd.build).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libdccf_hip.so is missing (%s): run `python -m dccf_amd.build` — there is no CPU fallback'
                           % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.dccf_last_error.restype = C.c_char_p
    lib.dccf_abi_version.restype = C.c_int
    if lib.dccf_abi_version() != ABI_VERSION:      # the ctypes structs below mirror include/dccf_hip.h of exactly this version
        raise RuntimeError('libdccf_hip.so has ABI version %d, this package needs %d: rebuild with `python -m dccf_amd.build`'
                           % (lib.dccf_abi_version(), ABI_VERSION))
    i64, i32, u64, f32, vp = C.c_int64, C.c_int32, C.c_uint64, C.c_float, C.c_void_p
    sig = {
        'dccf_ctx_create': [C.POINTER(vp), C.c_int],
        'dccf_ctx_destroy': [vp],
        'dccf_ctx_set_deterministic': [vp, C.c_int],
        'dccf_ctx_reserve': [vp, i64, i32, i32, i32, i32],
        'dccf_profile': [vp, C.c_int],
        'dccf_profile_read': [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
        'dccf_predict': [vp, C.POINTER(ModelT), C.POINTER(RandT), vp, i64, f32, vp, vp],
        'dccf_train_fwdbwd': [vp, C.POINTER(ModelT), C.POINTER(RandT), vp, vp, i64, i32, f32, C.POINTER(GradsT), vp, vp, vp],
        'dccf_dense_opt_step': [i32, vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, i32, vp],
        'dccf_dense_opt_step_rows': [i32, vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, i32, C.POINTER(i64), C.POINTER(i64),
                                     C.POINTER(i32), C.POINTER(vp), vp],
        'dccf_dense_opt_step_dev': [i32, vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, vp, i32, C.POINTER(i64), C.POINTER(i64),
                                    C.POINTER(i32), C.POINTER(vp), vp],
        'dccf_advance': [vp, vp],
        'dccf_dp_local': [vp, C.POINTER(ModelT), C.POINTER(RandT), vp, vp, i64, f32, C.POINTER(GradsT), C.POINTER(OptT),
                          C.POINTER(DpT), vp, u64, i32, vp, vp],
        'dccf_dp_overlap': [C.POINTER(OptT), C.POINTER(DpT), i32, C.POINTER(DpNextT), vp],
        'dccf_dp_finish': [C.POINTER(OptT), C.POINTER(DpT), i32, i32, C.POINTER(DpNextT), vp],
        'dccf_eval_prepare': [vp, C.POINTER(ModelT), vp, vp, vp],
        'dccf_predict_projected': [vp, C.POINTER(ModelT), C.POINTER(RandT), vp, i64, f32, vp, vp, vp, vp],
        'dccf_sample_eval_negatives': [vp, i64, vp, vp, i64, i32, u64, u64, vp, vp],
        'dp_import_apply': [vp, i32, i32, vp, vp, vp, i64, f32, f32, f32, f32, i64, i32, vp, vp, vp, vp, i64, vp, i64, i32, vp,
                            vp, vp, vp],
        'dccf_ctx_side_stream': [vp, C.POINTER(vp)],
        'dp_mark_global': [vp, i32, i64, i32, i64, u64, u64, vp, vp, i32, i32, vp, vp, vp, vp],
        'dccf_dense_opt_phase': [i32, vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, i32, C.POINTER(i64), C.POINTER(i64),
                                 C.POINTER(i32), C.POINTER(vp), i32, vp, vp, i64, vp],
        'dp_export_touched': [vp, i64, i32, vp, vp, vp, vp, i64, vp, vp, i64, i32, i32, vp],
        'dp_import_touched': [vp, i32, vp, i64, i32, vp, vp, vp, vp, i64, vp, i64, i32, vp, vp, vp, vp],
        'dccf_train_step': [vp, C.POINTER(ModelT), C.POINTER(RandT), vp, vp, i64, i32, f32, C.POINTER(GradsT),
                            C.POINTER(OptT), vp, vp, vp, u64, vp],
        'rank_eval_topk': [vp, vp, vp, vp, i64, C.POINTER(i32), vp, i32, vp, vp],
        'dccf_sumsq': [vp, i64, vp, vp],
        'mf_predict': [C.POINTER(MFModelT), vp, i64, vp, vp],
        'mf_train_fwdbwd': [vp, C.POINTER(MFModelT), vp, vp, i64, i32, C.POINTER(MFGradsT), vp, vp, vp],
        'mf_predict_full': [C.POINTER(MFModelT), vp, vp],
        'mf_train_step': [vp, C.POINTER(MFModelT), vp, vp, i64, i32, C.POINTER(MFGradsT), C.POINTER(OptT), vp, vp, vp, vp],
        'dccf_sample_train_negatives': [vp, vp, vp, vp, i64, i64, u64, u64, vp, vp],
        'dccf_ctx_prepared_steps': [vp, C.POINTER(C.c_int64)],
        'dccf_ctx_hosted_rows': [vp, C.POINTER(C.c_int64)],
        'dccf_debug_candidates': [i64, i32, i64, u64, u64, vp, vp],
        'dccf_debug_noise': [i64, i32, f32, u64, u64, vp, vp],
        'dccf_debug_keep': [i64, i32, f32, u64, u64, vp, vp],
        'dccf_debug_opt_elem': [i32, i32, vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, vp],
        'dccf_debug_keep_layer': [i64, i32, f32, u64, u64, i32, vp, vp],
        'dccf_debug_workspace': [vp, i64, i32, i32, i32, i32, i32, vp, C.POINTER(C.c_int64), vp],
        'shard_pack_rows': [vp, vp, i64, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, vp],
        'shard_unpack_rows': [vp, i32, i64, vp, C.POINTER(vp), C.POINTER(i32), i32, vp],
        'shard_scatter_add': [vp, i64, vp, i32, vp, vp, vp],
        'dccf_build_epoch_batches': [vp, vp, vp, vp, i64, i64, vp, vp, vp, u64, u64, vp],
        'shard_pack_multi': [vp, i32, vp],
        'dccf_lazy_scalars': [f32, i64, i32, vp],
        'dccf_lazy_catchup_rows': [C.POINTER(OptT), vp, i64, i32, vp, i64, i32, vp],
        'dccf_lazy_opt_step': [C.POINTER(OptT), i64, vp],
        'dccf_comm_unique_id': [vp],
        'dccf_comm_create': [C.POINTER(vp), vp, i32, i32],
        'dccf_comm_destroy': [vp],
        'dccf_comm_all_to_all_rows': [vp, vp, vp, vp, vp, i64, vp],
        'dccf_comm_all_to_all_rows2': [vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp],
        'dccf_comm_all_reduce_sum': [vp, vp, i64, vp],
        'dccf_lazy_flush': [C.POINTER(OptT), vp],
        'shard_unpack_multi': [vp, i32, vp, i64, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise RuntimeError('libdccf_hip: %s (code %d)' % (load().dccf_last_error().decode(), code))


def ptr(t, dtype=None):
    """Device address of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('expected a tensor in HBM (cuda device), got %s' % t.device)
    if not t.is_contiguous():
        raise RuntimeError('expected a contiguous tensor')
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError('expected dtype %s, got %s' % (dtype, t.dtype))
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """torch's current stream of the current device as a hipStream_t.  (torch.cuda.current_stream() builds a Stream object
    through several Python layers: 4-8 us per call, and a step of the sharded trainer asks ten times.)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class Context(object):
    """Opaque workspace owner (dccf_ctx)."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        check(load().dccf_ctx_create(C.byref(self.h), int(device)))

    def set_deterministic(self, on=True):
        """Deterministic gradient scatter for tests (dccf_ctx_set_deterministic): no float atomics in the backward, so every form of
        the training step gives the same bits."""
        check(load().dccf_ctx_set_deterministic(self.h, 1 if on else 0))

    def reserve(self, max_rows, D, F, S, A):
        check(load().dccf_ctx_reserve(self.h, int(max_rows), int(D), int(F), int(S), int(A)))

    def prepared_steps(self):
        """How many train steps started from a step the previous call prepared (dccf_ctx_prepared_steps)."""
        n = C.c_int64()
        check(load().dccf_ctx_prepared_steps(self.h, C.byref(n)))
        return n.value

    def hosted_rows(self):
        """Item rows whose untouched-row optimizer pass rode in the backward launch of the last training call."""
        n = C.c_int64()
        check(load().dccf_ctx_hosted_rows(self.h, C.byref(n)))
        return n.value

    def side_stream(self):
        """The context's low-priority (or CU-masked, DCCF_SIDE_CUS) stream as a torch stream."""
        h = C.c_void_p()
        check(load().dccf_ctx_side_stream(self.h, C.byref(h)))
        return torch.cuda.ExternalStream(h.value)

    KERNELS = ['prep', 'mlp_fwd', 'noise_fwd', 'pair_epilogue', 'mlp_bwd', 'noise_bwd_eps', 'opt_launch', 'lazy_catchup']

    def profile(self, enable=True):
        check(load().dccf_profile(self.h, 1 if enable else 0))

    def profile_read(self):
        """{kernel: (total ms, launches)} since the last read (synchronises the recorded events)."""
        ms, cnt = (C.c_double * 8)(), (C.c_int64 * 8)()
        check(load().dccf_profile_read(self.h, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(self.KERNELS) if cnt[i]}

    def __del__(self):
        try:
            if self.h:
                load().dccf_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass


def model_struct(U, V, W, b, feat, expo, S, A, std, ips=None, extra=None, expo_gathered=None):
    """extra: [(mlp.k.weight [D, D], mlp.k.bias [D]) for k = 1 .. n_layers - 1] (src/models/DCCF.py:61-62).
    expo_gathered: [N, S + 1] exposures of the call's own (row, candidate) slots instead of a matrix / factors."""
    m = ModelT()
    m.user_num, m.D = U.shape
    m.item_num = V.shape[0]
    m.F = feat.shape[1]
    m.S, m.A, m.std = int(S), int(A), float(std)
    f32 = torch.float32
    m.U, m.V, m.W, m.b, m.feat = ptr(U, f32), ptr(V, f32), ptr(W, f32), ptr(b, f32), ptr(feat, f32)
    m.expo = ptr(expo, f32)
    if tuple(W.shape) != (m.D, m.D + m.F) or V.shape[1] != m.D or feat.shape[0] != m.item_num:
        raise RuntimeError('inconsistent parameter shapes')
    if expo is not None and tuple(expo.shape) != (m.user_num, m.item_num):
        raise RuntimeError('expo must be [user_num, item_num]')
    if ips is not None:
        m.ipsP, m.ipsQ = ptr(ips['P'], f32), ptr(ips['Q'], f32)
        m.ipsBu, m.ipsBi, m.ipsProp = ptr(ips['bu'], f32), ptr(ips['bi'], f32), ptr(ips['prop'], f32)
        m.ipsB0, m.ipsM, m.ipsD = float(ips['b0']), float(ips['M']), int(ips['P'].shape[1])
    extra = list(extra or [])
    if len(extra) > 7:
        raise RuntimeError('at most 8 mlp layers (n_layers <= 8)')
    m.n_extra = len(extra)
    for k, (Wk, bk) in enumerate(extra):
        if tuple(Wk.shape) != (m.D, m.D) or tuple(bk.shape) != (m.D,):
            raise RuntimeError('extra mlp layers must be [D, D] / [D]')
        m.Wl[k], m.bl[k] = ptr(Wk, f32), ptr(bk, f32)
    m.expo_gathered = ptr(expo_gathered, f32)
    m._refs = (U, V, W, b, feat, expo, ips, extra, expo_gathered)   # the struct holds raw addresses: keep the tensors alive with it
    return m


def grads_struct(gU, gV, gW, gb, touchedU=None, touchedV=None, extra=None):
    """extra: [(grad of mlp.k.weight, grad of mlp.k.bias) for k = 1 .. n_layers - 1]."""
    g = GradsT(ptr(gU), ptr(gV), ptr(gW), ptr(gb), ptr(touchedU, torch.uint8), ptr(touchedV, torch.uint8))
    for k, (a, b) in enumerate(extra or []):
        g.gWl[k], g.gbl[k] = ptr(a, torch.float32), ptr(b, torch.float32)
    return g


def rand_struct(sample_item=None, noise=None, keep=None, seed=None, step=0, k_dev=None, x_stride=0, x_steps=1):
    r = RandT()
    if k_dev is not None:
        r.k_dev, r.x_stride, r.x_steps = ptr(k_dev, torch.int64), int(x_stride), int(x_steps)
        r._kref = k_dev
    if seed is not None and sample_item is not None:      # candidates injected, noise / dropout fused
        r.mode, r.seed, r.step = 2, int(seed) & 0xFFFFFFFFFFFFFFFF, int(step)
        r.sample_item = ptr(sample_item, torch.int64)
        r._refs = (sample_item,)
    elif seed is not None:
        r.mode, r.seed, r.step = 1, int(seed) & 0xFFFFFFFFFFFFFFFF, int(step)
    else:
        r.mode = 0
        r.sample_item = ptr(sample_item, torch.int64)
        r.noise = ptr(noise, torch.float32)
        r.keep = ptr(keep, torch.uint8)
        r._refs = (sample_item, noise, keep)
    return r


def dccf_predict(ctx, m, r, X, dropout, out=None):
    N = X.shape[0]
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    check(load().dccf_predict(ctx.h, C.byref(m), C.byref(r), ptr(X, torch.int64), N, float(dropout), ptr(out), stream()))
    return out


def dccf_train_fwdbwd(ctx, m, r, X, Y, rank, dropout, gU, gV, gW, gb, pred=None, loss=None, touchedU=None, touchedV=None,
                      gextra=None):
    N = X.shape[0]
    if pred is None:
        pred = torch.empty(N, dtype=torch.float32, device=X.device)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=X.device)
    g = grads_struct(gU, gV, gW, gb, touchedU, touchedV, gextra)
    check(load().dccf_train_fwdbwd(ctx.h, C.byref(m), C.byref(r), ptr(X, torch.int64), ptr(Y), N, int(rank),
                                   float(dropout), C.byref(g), ptr(pred), ptr(loss), stream()))
    return pred, loss


def opt_struct(kind, p, g, s1, s2, lr, wd, l2, clip, segments, overlap):
    """dccf_opt_t for dccf_train_step; `step` is filled in per call.  Keeps the host arrays and tensors alive."""
    n = len(segments)
    o = OptT()
    o._refs = [(C.c_int64 * max(n, 1))(*[int(s[0]) for s in segments]), (C.c_int64 * max(n, 1))(*[int(s[1]) for s in segments]),
               (C.c_int32 * max(n, 1))(*[int(s[2]) for s in segments]),
               (C.c_void_p * max(n, 1))(*[ptr(s[3], torch.uint8) for s in segments]), p, g, s1, s2, segments]
    o.kind, o.overlap = OPT_KIND[kind.lower()], int(overlap)
    o.p, o.g, o.s1, o.s2, o.n = ptr(p, torch.float32), ptr(g, torch.float32), ptr(s1), ptr(s2), p.numel()
    o.lr, o.wd, o.l2, o.clip, o.nseg = float(lr), float(wd), float(l2), float(clip), n
    o.seg_begin, o.seg_rows, o.seg_width, o.seg_flags = [C.cast(a, C.c_void_p) for a in o._refs[:4]]
    return o


class Comm(object):
    """The row-sharded trainer's collectives straight on RCCL (include/dccf_hip.h, dccf_comm_*): one C call each on torch's
    current stream instead of a torch.distributed call.  Collective constructor: every rank of `group` calls it."""

    def __init__(self, rank, world, device, group=None):
        import torch.distributed as dist
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            check(load().dccf_comm_unique_id(buf))
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        ident = ident.to(device)
        if world > 1:
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = (C.c_uint8 * 128)(*ident.cpu().tolist())
        self.h = C.c_void_p()
        self.world = int(world)
        with torch.cuda.device(device):
            check(load().dccf_comm_create(C.byref(self.h), raw, int(world), int(rank)))

    def all_to_all_rows(self, out, inp, send_rows, recv_rows, width):
        """send_rows / recv_rows: HOST addresses (int) of int64[world] arrays, rows per peer — the caller keeps the arrays alive
        and does the address arithmetic (numpy's .ctypes.data costs microseconds per call); out / inp: float32 tensors."""
        check(load().dccf_comm_all_to_all_rows(self.h, inp.data_ptr(), send_rows, out.data_ptr(), recv_rows, width, stream()))

    def all_to_all_rows2(self, out_a, inp_a, send_a, recv_a, out_b, inp_b, send_b, recv_b):
        """Two payloads of different row widths in one RCCL group."""
        check(load().dccf_comm_all_to_all_rows2(self.h, inp_a.data_ptr(), send_a, out_a.data_ptr(), recv_a, out_a.shape[1],
                                                inp_b.data_ptr(), send_b, out_b.data_ptr(), recv_b, out_b.shape[1], stream()))

    def all_reduce_sum(self, buf):
        check(load().dccf_comm_all_reduce_sum(self.h, ptr(buf), buf.numel(), stream()))

    def close(self):
        if self.h:
            load().dccf_comm_destroy(self.h)
            self.h = C.c_void_p()


class LazyState(object):
    """The arrays of the windowed lazy regularisation of a dccf_opt_t (include/dccf_hip.h), owned here as torch tensors: per-row
    step counters, claims, the step's row list, and the table of Adam step scalars (refilled when the step runs off its end)."""
    NSCAL = 1 << 15
    _ids = 0

    def __init__(self, opt, K, n_rows, list_cap, lr, device):
        self.opt, self.K, self.lr = opt, int(K), float(lr)
        self.last = torch.zeros(n_rows, dtype=torch.int32, device=device)
        self.claim = torch.zeros(2 * n_rows, dtype=torch.int32, device=device)        # by step parity (include/dccf_hip.h)
        self.list_cap = int(list_cap)
        self.list = torch.zeros(2 * self.list_cap, dtype=torch.int32, device=device)
        self.cnt = torch.zeros(16, dtype=torch.int32, device=device)      # the pending-window records (include/dccf_hip.h)
        self.scal = torch.zeros(4 * self.NSCAL, dtype=torch.float32, device=device)
        self.t0 = -1
        self.dirty = False          # True while some row may be behind opt.step
        # the last step whose lazy optimizer launch went out (dccf_opt_t.lazy_host: the library checks that steps arrive one by one)
        self.host = (C.c_int64 * 2)(0, 0)
        opt.lazy_host = C.addressof(self.host)
        opt.lazy_K, opt.lazy_nscal = self.K, self.NSCAL
        opt.lazy_last, opt.lazy_claim = ptr(self.last, torch.int32), ptr(self.claim, torch.int32)
        opt.lazy_list, opt.lazy_cnt = ptr(self.list, torch.int32), ptr(self.cnt, torch.int32)
        opt.lazy_scal = ptr(self.scal, torch.float32)
        opt.lazy_list_cap = self.list_cap
        LazyState._ids += 1
        opt.lazy_id = LazyState._ids

    def step_list(self, step, n):
        """The first n slots of the row list of optimizer step `step`."""
        o = (int(step) & 1) * self.list_cap
        return self.list[o:o + n]

    def cover(self, step):
        """Makes the scalar table cover [step - K + 1, step] (and a long stretch beyond)."""
        lo = max(1, step - self.K + 1)
        if self.t0 < 0 or lo < self.t0 or step >= self.t0 + self.NSCAL:
            self.t0 = max(0, lo - 1)
            host = (C.c_float * (4 * self.NSCAL))()
            check(load().dccf_lazy_scalars(self.lr, self.t0, self.NSCAL, host))
            self.scal.copy_(torch.frombuffer(host, dtype=torch.float32).clone().to(self.scal.device))
            self.opt.lazy_t0 = self.t0

    def sync_all(self, step):
        """Every row IS at `step` (dense steps ran, or nothing ran yet): reset the counters."""
        self.last.fill_(int(step))
        self.host[0] = int(step)
        self.dirty = False

    def catchup_rows(self, step, rows, n, seg=0):
        """Before a step that drives its own launches (the row-sharded trainer) reads the parameters: rows[:n] (int32, of segment
        `seg`) are claimed for `step`, listed and brought up to step - 1 (dccf_lazy_catchup_rows)."""
        self.opt.step = int(step)
        self.cover(step)
        self.dirty = True
        check(load().dccf_lazy_catchup_rows(C.byref(self.opt), ptr(rows, torch.int32), int(n), int(seg), None, 0, 0, stream()))

    def opt_step(self, step, nslots):
        """The optimizer launch of that step (dccf_lazy_opt_step), once the gradient rows are complete."""
        self.opt.step = int(step)
        check(load().dccf_lazy_opt_step(C.byref(self.opt), int(nslots), stream()))

    def flush(self, step):
        """Brings every row up to `step` (dccf_lazy_flush); no-op when nothing is behind."""
        if not self.dirty:
            return
        self.opt.step = int(step)
        self.cover(step)
        check(load().dccf_lazy_flush(C.byref(self.opt), stream()))
        self.dirty = False
        self.check_rows()

    def check_rows(self):
        """Raises if a kernel met a row more than K steps behind (lazy_cnt[15]: steps were skipped behind the library's back).
        Synchronises; called after a flush, i.e. before evaluation / checkpoints, never inside the step loop."""
        if int(self.cnt[15]) != 0:
            raise RuntimeError('lazy optimizer: a parameter row was more than lazy_K steps behind — steps did not arrive one by one')


def dccf_train_step(ctx, m, r, X, Y, rank, dropout, gU, gV, gW, gb, opt, step, pred=None, loss=None, touchedU=None,
                    touchedV=None, X_next=None, step_next=0, gextra=None):
    """forward + loss + backward + regularised optimizer step (dccf_train_step): BaseRunner.py:172-188 in one call.
    X_next (same shape as X) = the next call's batch, whose Philox step will be step_next: prepared inside this call's
    optimizer launch."""
    N = X.shape[0]
    if X_next is not None and tuple(X_next.shape) != tuple(X.shape):
        X_next = None
    if pred is None:
        pred = torch.empty(N, dtype=torch.float32, device=X.device)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=X.device)
    g = grads_struct(gU, gV, gW, gb, touchedU, touchedV, gextra)
    opt.step = int(step)
    check(load().dccf_train_step(ctx.h, C.byref(m), C.byref(r), ptr(X, torch.int64), ptr(Y), N, int(rank), float(dropout),
                                 C.byref(g), C.byref(opt), ptr(pred), ptr(loss), ptr(X_next, torch.int64), int(step_next),
                                 stream()))
    return pred, loss


def dense_opt_step(kind, p, g, s1, s2, lr, wd, l2, clip, step, zero_grad=True):
    check(load().dccf_dense_opt_step(OPT_KIND[kind.lower()], ptr(p, torch.float32), ptr(g, torch.float32), ptr(s1),
                                     ptr(s2), p.numel(), float(lr), float(wd), float(l2), float(clip), int(step),
                                     1 if zero_grad else 0, stream()))


def dense_opt_step_rows(kind, p, g, s1, s2, lr, wd, l2, clip, step, segments, k_dev=None):
    """segments: list of (begin element, rows, row width, touched uint8 tensor) — see dccf_dense_opt_step_rows.
    With k_dev (int64 device scalar) the step is step + *k_dev (graph-replayable form)."""
    n = len(segments)
    beg = (C.c_int64 * max(n, 1))(*[int(s[0]) for s in segments])
    rows = (C.c_int64 * max(n, 1))(*[int(s[1]) for s in segments])
    wid = (C.c_int32 * max(n, 1))(*[int(s[2]) for s in segments])
    fl = (C.c_void_p * max(n, 1))(*[ptr(s[3], torch.uint8) for s in segments])
    a = (OPT_KIND[kind.lower()], ptr(p, torch.float32), ptr(g, torch.float32), ptr(s1), ptr(s2), p.numel(), float(lr),
         float(wd), float(l2), float(clip), int(step))
    if k_dev is None:
        check(load().dccf_dense_opt_step_rows(*a, n, beg, rows, wid, fl, stream()))
    else:
        check(load().dccf_dense_opt_step_dev(*a, ptr(k_dev, torch.int64), n, beg, rows, wid, fl, stream()))


def advance(k_dev):
    check(load().dccf_advance(ptr(k_dev, torch.int64), stream()))


def sumsq(p):
    out = torch.zeros(1, dtype=torch.float32, device=p.device)
    check(load().dccf_sumsq(ptr(p, torch.float32), p.numel(), ptr(out), stream()))
    return out


def mf_struct(kind, P, Q, bu=None, bi=None, b0=None, prop=None, M=0.1):
    m = MFModelT()
    m.user_num, m.D = P.shape
    m.item_num = Q.shape[0]
    m.kind = MF_KIND[kind]
    f32 = torch.float32
    m.P, m.Q, m.bu, m.bi, m.b0, m.prop = ptr(P, f32), ptr(Q, f32), ptr(bu, f32), ptr(bi, f32), ptr(b0, f32), ptr(prop, f32)
    m.M = float(M)
    m._refs = (P, Q, bu, bi, b0, prop)
    return m


def mf_predict(m, X, out=None):
    N = X.shape[0]
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    check(load().mf_predict(C.byref(m), ptr(X, torch.int64), N, ptr(out), stream()))
    return out


def mf_train_fwdbwd(ctx, m, X, Y, rank, gP, gQ, gbu=None, gbi=None, gb0=None, pred=None, loss=None, touchedP=None,
                    touchedQ=None):
    N = X.shape[0]
    if pred is None:
        pred = torch.empty(N, dtype=torch.float32, device=X.device)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=X.device)
    g = MFGradsT(ptr(gP), ptr(gQ), ptr(gbu), ptr(gbi), ptr(gb0), ptr(touchedP, torch.uint8), ptr(touchedQ, torch.uint8))
    check(load().mf_train_fwdbwd(ctx.h if ctx is not None else None, C.byref(m), ptr(X, torch.int64), ptr(Y), N,
                                 int(rank), C.byref(g), ptr(pred), ptr(loss), stream()))
    return pred, loss


def mf_train_step(ctx, m, X, Y, rank, gP, gQ, gbu, gbi, gb0, opt, step, ids, pred=None, loss=None):
    """One MF train step under the lazy optimizer (mf_train_step): catch-up of the batch's rows, forward + loss + backward, the
    optimizer launch of step `step`.  ids: int32 scratch [2 N]."""
    N = X.shape[0]
    if pred is None:
        pred = torch.empty(N, dtype=torch.float32, device=X.device)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=X.device)
    g = MFGradsT(ptr(gP), ptr(gQ), ptr(gbu), ptr(gbi), ptr(gb0), None, None)
    opt.step = int(step)
    check(load().mf_train_step(ctx.h if ctx is not None else None, C.byref(m), ptr(X, torch.int64), ptr(Y), N, int(rank),
                               C.byref(g), C.byref(opt), ptr(ids, torch.int32), ptr(pred), ptr(loss), stream()))
    return pred, loss


def mf_predict_full(m, out=None, device=None):
    if out is None:
        out = torch.empty((m.user_num, m.item_num), dtype=torch.float32, device=device)
    check(load().mf_predict_full(C.byref(m), ptr(out, torch.float32), stream()))
    return out


def sample_train_negatives(rows_indptr, rows, hist_indptr, hist_items, user_num, item_num, seed, epoch, out=None):
    if out is None:
        out = torch.empty(rows.shape[0], dtype=torch.int64, device=rows.device)
    i64 = torch.int64
    check(load().dccf_sample_train_negatives(ptr(rows_indptr, i64), ptr(rows, i64), ptr(hist_indptr, i64),
                                             ptr(hist_items, i64), int(user_num), int(item_num),
                                             int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch), ptr(out, i64), stream()))
    return out


def debug_candidates(N, S, item_num, seed, step, device):
    out = torch.empty((N, S), dtype=torch.int64, device=device)
    check(load().dccf_debug_candidates(N, S, item_num, int(seed), int(step), ptr(out), stream()))
    return out


def debug_noise(L, F, std, seed, step, device):
    out = torch.zeros((L, F), dtype=torch.float32, device=device)
    check(load().dccf_debug_noise(L, F, float(std), int(seed), int(step), ptr(out), stream()))
    return out


def debug_opt_elem(kind, ieee, p, g, s1, s2, lr, wd, l2, clip, step):
    """One optimizer step, element by element, in place: the library's element function (ieee=0) or its IEEE twin (ieee=1)."""
    o = lambda t: ptr(t) if t is not None else None
    check(load().dccf_debug_opt_elem(OPT_KIND[kind], int(ieee), ptr(p), ptr(g), o(s1), o(s2), p.numel(), float(lr), float(wd),
                                     float(l2), float(clip), int(step), stream()))


def debug_keep(L, D, dropout, seed, step, device, layer=0):
    out = torch.empty((L, D), dtype=torch.uint8, device=device)
    check(load().dccf_debug_keep_layer(L, D, float(dropout), int(seed), int(step), int(layer), ptr(out), stream()))
    return out


def debug_workspace(ctx, N, D, F, S, A, which, device):
    """Workspace array `which` of the last call (tests only): 0 cand 1 WT 3 h 4 m 5 dmns."""
    info = (C.c_int64 * 4)()
    check(load().dccf_debug_workspace(ctx.h, N, D, F, S, A, which, None, info, stream()))
    out = torch.empty(info[2], dtype=torch.int32 if which == 0 else torch.float32, device=device)
    check(load().dccf_debug_workspace(ctx.h, N, D, F, S, A, which, ptr(out), info, stream()))
    return out, int(info[0]), int(info[1])


def _table_args(tables):
    n = len(tables)
    ptrs = (C.c_void_p * n)(*[ptr(t, torch.float32) for t in tables])
    widths = (C.c_int32 * n)(*[int(t.shape[1]) if t.dim() == 2 else 1 for t in tables])
    return ptrs, widths, n


def shard_pack_rows(idx, dst, n, tables, out):
    """out[dst[j] or j, :w] = [T0[idx[j]] | T1[idx[j]] | ...] for j < n (idx, dst int32 in HBM; out rows may be wider)."""
    ptrs, widths, k = _table_args(tables)
    check(load().shard_pack_rows(ptr(idx, torch.int32), ptr(dst, torch.int32), int(n), ptrs, widths, k,
                                 ptr(out, torch.float32), int(out.shape[1]), stream()))


def shard_unpack_rows(payload, n, dst, tables):
    ptrs, widths, k = _table_args(tables)
    check(load().shard_unpack_rows(ptr(payload, torch.float32), int(payload.shape[1]), int(n), ptr(dst, torch.int32), ptrs,
                                   widths, k, stream()))


def build_epoch_batches(uid, iid, neg, perm, batch_size, bad, seed=0, epoch=0):
    """(full [n // B, 2B, 2], tail [2 (n % B), 2] or None) int64 — one launch.  perm: the epoch's permutation as an int64
    tensor, or None: a keyed bijection of (seed, epoch) computed inside the kernel."""
    n = uid.numel()
    nb, r = n // batch_size, n % batch_size
    full = torch.empty((nb, 2 * batch_size, 2), dtype=torch.int64, device=uid.device)
    tail = torch.empty((2 * r, 2), dtype=torch.int64, device=uid.device) if r else None
    check(load().dccf_build_epoch_batches(ptr(uid, torch.int64), ptr(iid, torch.int64), ptr(neg, torch.int64), ptr(perm, torch.int64),
                                          n, int(batch_size), ptr(full, torch.int64), ptr(tail, torch.int64), ptr(bad, torch.int32),
                                          int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch) & 0xFFFFFFFFFFFFFFFF, stream()))
    return full, tail


def shard_jobs(jobs):
    """A HOST array of shard_job_t from [(idx, dst, n, tables, payload[, col])] (idx / dst int32 tensors or None; col: the job's
    rows start at column `col` of the payload rows); keeps the tensors alive.  `n`, `idx` and `dst` may be rewritten per step
    through the returned array (a[q].n = ..., a[q].idx = ptr)."""
    a = (ShardJobT * len(jobs))()
    keep = []
    for q, job in enumerate(jobs):
        idx, dst, n, tables, payload = job[:5]
        col = int(job[5]) if len(job) > 5 else 0
        a[q].idx, a[q].dst, a[q].n = ptr(idx, torch.int32), ptr(dst, torch.int32), int(n)
        width = 0
        for t, tb in enumerate(tables):
            a[q].tables[t] = ptr(tb, torch.float32)
            a[q].widths[t] = int(tb.shape[1]) if tb.dim() > 1 else 1
            width += a[q].widths[t]
        if col < 0 or col + width > int(payload.shape[1]):
            raise RuntimeError('shard job: columns [%d, %d) do not fit the payload rows (%d floats)' % (col, col + width, payload.shape[1]))
        a[q].ntables, a[q].ld, a[q].buf = len(tables), int(payload.shape[1]), ptr(payload, torch.float32) + 4 * col
        keep.append((idx, dst, tables, payload))
    a._keep = keep
    return a


def shard_pack_multi(jobs):
    check(load().shard_pack_multi(C.cast(jobs, C.c_void_p), len(jobs), stream()))


def shard_unpack_multi(jobs, zero=None):
    check(load().shard_unpack_multi(C.cast(jobs, C.c_void_p), len(jobs), ptr(zero, torch.float32), zero.numel() if zero is not None else 0,
                                    stream()))


def shard_scatter_add(idx, n, rows, g, flags=None):
    check(load().shard_scatter_add(ptr(idx, torch.int32), int(n), ptr(rows, torch.float32), int(g.shape[1]),
                                   ptr(g, torch.float32), ptr(flags, torch.uint8), stream()))


def rank_eval_topk(pred, label, indptr, rows, ks):
    """Per-user (ndcg, hit, precision, recall)@k for up to 4 cut-offs k <= 1024 -> tensor [n_users, len(ks)+1, 4]."""
    nu = indptr.numel() - 1
    nk = len(ks)
    ks_host = (C.c_int32 * nk)(*[int(k) for k in ks])
    ks_dev = torch.tensor([int(k) for k in ks], dtype=torch.int32, device=pred.device)
    out = torch.zeros((nu, nk + 1, 4), dtype=torch.float32, device=pred.device)
    check(load().rank_eval_topk(ptr(pred, torch.float32), ptr(label, torch.float32), ptr(indptr, torch.int64),
                                ptr(rows, torch.int64), nu, ks_host, ptr(ks_dev, torch.int32), nk, ptr(out), stream()))
    return out


def _seg_arrays(segments):
    n = len(segments)
    return (n, (C.c_int64 * n)(*[int(s[0]) for s in segments]), (C.c_int64 * n)(*[int(s[1]) for s in segments]),
            (C.c_int32 * n)(*[int(s[2]) for s in segments]), (C.c_void_p * n)(*[ptr(s[3], torch.uint8) for s in segments]))


def dp_buffer_words(cap, D, nd):
    f = load().dp_buffer_words
    f.restype, f.argtypes = C.c_int64, [C.c_int64, C.c_int32, C.c_int64]
    return int(f(int(cap), int(D), int(nd)))


def dp_export_touched(g, segments, dense_begin, loss, buf, cap, D, reset=True):
    """Moves the touched gradient rows + the dense tail of flat `g` into this rank's all-gather buffer."""
    n, beg, rows, wid, fl = _seg_arrays(segments)
    check(load().dp_export_touched(ptr(g, torch.float32), g.numel(), n, beg, rows, wid, fl, int(dense_begin), ptr(loss),
                                   ptr(buf, torch.float32), int(cap), int(D), 1 if reset else 0, stream()))


class DpScratch(object):
    """Scratch of dp_import_touched for R rows and G ranks."""

    def __init__(self, R, G, device):
        self.mask = torch.zeros(R, dtype=torch.int32, device=device)
        self.where = torch.empty(G * R, dtype=torch.int32, device=device)


def dp_import_touched(bufs, G, g, segments, dense_begin, loss_sum, cap, D, scratch, reset_buf=None):
    """g <- rank-ordered sum of the G gathered buffers (rows, dense tail); sets the touched bytes."""
    n, beg, rows, wid, fl = _seg_arrays(segments)
    check(load().dp_import_touched(ptr(bufs, torch.float32), int(G), ptr(g, torch.float32), g.numel(), n, beg, rows, wid, fl,
                                   int(dense_begin), ptr(loss_sum), int(cap), int(D), ptr(scratch.mask, torch.int32),
                                   ptr(scratch.where, torch.int32), ptr(reset_buf), stream()))


def dp_mark_global(X_all, S, item_num, seed, step0, flagsU, flagsV, list_, cnt, parity, segU=0, segV=1):
    """Marks every row any rank touches in this step (bytes + list); cnt: int32 [2] device counters, parity selects."""
    G, N = X_all.shape[0], X_all.shape[1]
    c = ptr(cnt, torch.int32)
    check(load().dp_mark_global(ptr(X_all, torch.int64), G, N, int(S), int(item_num), int(seed) & 0xFFFFFFFFFFFFFFFF, int(step0),
                                ptr(flagsU, torch.uint8), ptr(flagsV, torch.uint8), segU, segV, ptr(list_, torch.int64),
                                c + 4 * parity, c + 4 * (1 - parity), stream()))


def dense_opt_phase(kind, p, g, s1, s2, lr, wd, l2, clip, step, segments, phase, list_=None, cnt=None, parity=0, max_rows=0):
    n, beg, rows, wid, fl = _seg_arrays(segments)
    c = ptr(cnt, torch.int32)
    check(load().dccf_dense_opt_phase(OPT_KIND[kind.lower()], ptr(p, torch.float32), ptr(g, torch.float32), ptr(s1), ptr(s2),
                                      p.numel(), float(lr), float(wd), float(l2), float(clip), int(step), n, beg, rows, wid, fl,
                                      int(phase), ptr(list_, torch.int64), (c + 4 * parity) if c else None, int(max_rows),
                                      stream()))


def sample_eval_negatives(users, hist_indptr, hist_items, item_num, neg_n, seed, tag):
    """int64 [n_users, neg_n]: the eval negatives of every distinct user (see dccf_sample_eval_negatives)."""
    i64 = torch.int64
    out = torch.empty((users.shape[0], int(neg_n)), dtype=i64, device=users.device)
    check(load().dccf_sample_eval_negatives(ptr(users, i64), users.shape[0], ptr(hist_indptr, i64), ptr(hist_items, i64),
                                            int(item_num), int(neg_n), int(seed) & 0xFFFFFFFFFFFFFFFF, int(tag), ptr(out, i64),
                                            stream()))
    return out


def dccf_eval_prepare(ctx, m, Pf=None, Lt=None):
    """Pf [item_num, D] = feat W_f^T and Lt [D, D] (Cholesky factor of std^2 W_f W_f^T, transposed) for
    dccf_predict_projected; call after the parameters changed."""
    dev = m._refs[0].device
    if Pf is None:
        Pf = torch.empty((m.item_num, m.D), dtype=torch.float32, device=dev)
    if Lt is None:
        Lt = torch.empty((m.D, m.D), dtype=torch.float32, device=dev)
    check(load().dccf_eval_prepare(ctx.h, C.byref(m), ptr(Pf, torch.float32), ptr(Lt, torch.float32), stream()))
    return Pf, Lt


def dccf_predict_projected(ctx, m, r, X, dropout, Pf, Lt, out=None):
    N = X.shape[0]
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    check(load().dccf_predict_projected(ctx.h, C.byref(m), C.byref(r), ptr(X, torch.int64), N, float(dropout),
                                        ptr(Pf, torch.float32), ptr(Lt, torch.float32), ptr(out), stream()))
    return out
