# coding=utf-8
"""Replicated data-parallel DCCF training across the GPUs of one node (the default of ``bench.py --gpus N``).

The reference is single-GPU (src/main.py:106,153-155); this is new capability with no reference counterpart.

MI355X-first partitioning: nothing is sharded.  The whole model is 16.4 M parameters (66 MB; 197 MB with Adam's state),
the frozen tables are 193 MB (768-d features) + 48.5 GB (dense exposure matrix) at Electronics size — a 288 GB GPU holds
all of it, so every rank keeps a full replica and the per-step exchange shrinks to what a step PRODUCES:

  * each rank runs the unchanged single-GPU forward/backward kernels on its own ``batch_size`` pairs (one process per GPU,
    ``torch.distributed`` backend "nccl" = RCCL over xGMI); Philox step word = t * G + rank, so the ranks draw
    independent candidates / noise / dropout;
  * ``dp_export_touched`` compacts the ~3 k embedding rows the batch touched (row id + gradient row) and the dense
    ``[dW | db]`` (213 KB) into ONE buffer (~1 MB at batch 128);
  * ONE collective per step: ``all_gather_into_tensor`` of those buffers;
  * ``dp_import_touched`` sums the G buffers into the local flat gradient IN RANK ORDER (no float atomics), so every
    replica holds bit-identical gradients and, after the same dense regularised Adam pass, bit-identical parameters —
    replicas cannot drift, no parameter broadcast is ever needed.

Hiding the collective (``overlap``): the schedule and the candidate streams are replicated, so every rank knows BEFORE the
step which rows ANY rank will touch (``dp_mark_global``).  All other rows — 98 % of the tables — see only the l2 term: their
optimizer pass (``dccf_dense_opt_phase`` 1, the 58 us HBM-bound bulk of a step) is enqueued right after the all-gather was
handed to RCCL's stream and runs while the buffers travel; the listed rows + W, b follow after the import (phase 2).

One optimizer step therefore sees the sum of the G ranks' BPR terms: exactly the reference's step at batch size G * B
(loss = -sum log sigmoid, src/models/DCCF.py:116-120).  The row-sharded alternative (dccf_amd/sharded.py) is for tables
that do not fit one GPU; it needs 4 collectives per step instead of 1.
"""
import os

import torch
import torch.distributed as dist

# rehearsal on one GPU: issue the all-gather even at world size 1 (measures the host cost of the collective call)
_FORCE_COLLECTIVE = os.environ.get('DCCF_FORCE_COLLECTIVE', '0') == '1'


class HipBackend(object):
    """The product backend: libdccf_hip.so through dccf_amd._lib.  One step = three library calls (dccf_dp_local,
    dccf_dp_overlap, dccf_dp_finish) whose argument blocks are built once (`prepare`): a step at batch 128 is ~150 us of
    GPU work and the host must stay below that (measured: 7 separate calls + the collective cost 143 us of host time)."""

    def __init__(self, device):
        from dccf_amd import _lib
        self.L = _lib
        self.device = device
        self.ctx = _lib.Context(device.index or 0)
        self.ready = None

    def prepare(self, tr):
        import ctypes as C
        L = self.L
        lib = L.load()
        f32, u8, i32, i64 = torch.float32, torch.uint8, torch.int32, torch.int64
        self.m = L.model_struct(tr.U, tr.V, tr.W, tr.b, tr.feat, tr.expo, tr.S, tr.A, tr.std, ips=tr.ips, extra=tr.extra)
        self.r = L.rand_struct(seed=tr.seed, step=0)
        self.g = L.grads_struct(tr.gU, tr.gV, tr.gW, tr.gb, tr.tU, tr.tV, tr.gextra)
        self.opt = L.opt_struct(tr.opt_name, tr.flat_p, tr.flat_g, tr.s1, tr.s2, tr.lr, tr.l2, tr.l2, 50.0, tr.segments, 0)
        # windowed lazy regularisation in the overlapped form of the step: phase 1 advances one K-th of the
        # rows no rank touches by K steps instead of streaming all of them every step
        self.lazy = None
        if tr.lazy_K >= 2 and tr.overlap and tr.D % 4 == 0 and 4 <= tr.D <= 128:       # (any row width the lazy kernels take)
            self.lazy = L.LazyState(self.opt, tr.lazy_K, tr.user_num + tr.item_num, 64, tr.lr, tr.flat_p.device)
            self.lazy.sync_all(tr.t)
        self.scratch = L.DpScratch(tr.user_num + tr.item_num, tr.G, tr.flat_g.device)
        d = L.DpT()
        d.G, d.rank, d.D, d.S = int(tr.G), int(tr.rank), int(tr.D), int(tr.S)
        d.cap, d.dense_begin, d.item_num = int(tr.cap), int(tr.dense_begin), int(tr.item_num)
        d.seed = int(tr.seed) & 0xFFFFFFFFFFFFFFFF
        d.buf, d.bufs, d.loss, d.loss_sum = L.ptr(tr.buf, f32), L.ptr(tr.bufs, f32), L.ptr(tr.loss, f32), L.ptr(tr.loss_sum, f32)
        d.gflagsU, d.gflagsV, d.segU, d.segV = L.ptr(tr.gfU, u8), L.ptr(tr.gfV, u8), 0, 1
        d.gflagsU2, d.gflagsV2 = L.ptr(tr.gfU2, u8), L.ptr(tr.gfV2, u8)
        d.lflagsU, d.lflagsV, d.llist, d.lcnt = L.ptr(tr.lfU, u8), L.ptr(tr.lfV, u8), L.ptr(tr.llist, i64), L.ptr(tr.lcnt, i32)
        d.ctx, d.pmask, d.pwhere = self.ctx.h, L.ptr(tr.pmask, i32), L.ptr(tr.pwhere, i32)
        d.glist, d.gcnt = L.ptr(tr.glist, i64), L.ptr(tr.gcnt, i32)
        d.mask, d.where = L.ptr(self.scratch.mask, i32), L.ptr(self.scratch.where, i32)
        self.dp = d
        self.nx = L.DpNextT()
        self.nx.ctx, self.nx.model = self.ctx.h, C.pointer(self.m)
        self.nxp = C.byref(self.nx)
        self.mp, self.rp, self.gp, self.op, self.dpp = (C.byref(self.m), C.byref(self.r), C.byref(self.g), C.byref(self.opt),
                                                        C.byref(self.dp))
        self.f_local, self.f_overlap, self.f_finish = lib.dccf_dp_local, lib.dccf_dp_overlap, lib.dccf_dp_finish
        self.ready = tr

    def local(self, tr, X, Y, step, pred, X_all=None, step0=0):
        """forward/backward on this rank's batch + export of the touched gradient rows into tr.buf; with X_all the rows ANY
        rank touches this step are marked in the same launch (overlap mode)."""
        if self.ready is not tr:
            self.prepare(tr)
        if pred is None:
            pred = torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
        if self.lazy is not None:
            if X_all is None:                  # a step without the schedule: bring everything up to date, dense pass for this one
                self.flush(tr)
                self.opt.lazy_K = 0
            else:
                self.opt.lazy_K = self.lazy.K
                self.opt.step = tr.t + 1       # (the catch-up needs the step number before the forward)
                self.lazy.cover(tr.t + 1)
                self.lazy.dirty = True
        self.r.step = step
        self.L.check(self.f_local(self.ctx.h, self.mp, self.rp, self.L.ptr(X, torch.int64), self.L.ptr(Y, torch.float32),
                                  X.shape[0], tr.dropout, self.gp, self.op, self.dpp, self.L.ptr(X_all, torch.int64), int(step0),
                                  tr.parity, self.L.ptr(pred, torch.float32), self.L.stream()))
        return pred

    def _next(self, tr):
        """tr.next = (X_next, X_all_next) of the step after this one, or None: what the optimizer launches may prepare."""
        nxt = getattr(tr, 'next', None)
        if nxt is None:
            return None
        X_next, X_all_next = nxt
        nx = self.nx
        nx.X_next, nx.X_all_next = self.L.ptr(X_next, torch.int64), self.L.ptr(X_all_next, torch.int64)
        nx.N, nx.step0_next = X_next.shape[0], tr.word_base + tr.t * tr.G   # tr.t was already advanced: rank 0's next Philox step
        return self.nxp

    def overlap(self, tr, t):
        """While RCCL moves the buffers: the optimizer pass over the rows no rank touches (+ the next step's preparation)."""
        self.opt.step = t
        self.L.check(self.f_overlap(self.op, self.dpp, tr.parity, self._next(tr), self.L.stream()))

    def flush(self, tr):
        """Every row up to step tr.t (no-op unless lazy rows are behind): before anything but train_step reads the tables."""
        if self.lazy is not None and self.ready is tr:
            self.opt.lazy_K = self.lazy.K
            self.lazy.flush(tr.t)

    def finish(self, tr, t, ov):
        """After the gathered buffers arrived: rank-ordered sums -> optimizer (ov: only the marked rows + W, b are left)."""
        self.opt.step = t
        if self.lazy is not None and self.opt.lazy_K == 0:      # the dense step of a call without the schedule
            self.lazy.sync_all(int(t))
        self.L.check(self.f_finish(self.op, self.dpp, 1 if ov else 0, tr.parity, self._next(tr) if ov else None,
                                   self.L.stream()))


class ReplicatedDCCF(object):
    lazy_K = int(os.environ.get('DCCF_LAZY_K', '8'))       # 0: the dense pass in every phase 1

    def flush(self):
        """Brings every parameter row up to the current step (the lazy regularisation leaves rows behind between steps)."""
        if hasattr(self.be, 'flush'):
            self.be.flush(self)

    def __init__(self, rank, world, user_num, item_num, D, S, A, std, dropout, lr, l2, seed, backend, device, feat,
                 expo=None, ips=None, max_rows=256, group=None, overlap=True, flat_p=None, s1=None, s2=None, opt_name='adam', n_layers=1):
        """feat [item_num, F]; expo [user_num, item_num] or ips (dict of IPSBiasedMF factors) — full tables, identical on
        every rank.  max_rows: the largest 2B a step will see (sizes the all-gather buffer).  flat_p / s1 / s2: train THESE
        buffers (a models.DCCF's flat parameter buffer and its optimizer's Adam state: same [U | V | W | b] layout, every
        block on a 256-float boundary) instead of allocating — how runner.fit puts a CLI model on G GPUs.  opt_name: 'adam' |
        'adagrad' | 'gd' (src/runners/BaseRunner.py:83-107; s1 = Adam's m / Adagrad's sum, s2 = Adam's v).  n_layers: the
        extra Linear(D, D) layers of --n_layers > 1 (src/models/DCCF.py:61-62) follow b in the buffer, weight then bias per
        layer: they are part of the dense tail that travels and is summed with [dW | db]."""
        self.n_layers = int(n_layers)
        self.opt_name = opt_name.lower()
        if self.opt_name not in ('adam', 'adagrad', 'gd'):
            raise ValueError('unknown optimizer ' + opt_name)
        self.rank, self.G, self.group, self.dev, self.be = rank, world, group, device, backend
        self.user_num, self.item_num, self.D, self.S, self.A = user_num, item_num, D, S, A
        self.std, self.dropout, self.lr, self.l2, self.seed = std, dropout, lr, l2, seed
        self.feat, self.expo, self.ips = feat, expo, ips
        F = feat.shape[1]
        sizes = [user_num * D, item_num * D, D * (D + F), D] + [D * D, D] * (self.n_layers - 1)
        shapes = [(user_num, D), (item_num, D), (D, D + F), (D,)] + [(D, D), (D,)] * (self.n_layers - 1)
        pads = [(n + 255) // 256 * 256 for n in sizes]
        f32 = torch.float32
        for name, t in (('flat_p', flat_p), ('s1', s1), ('s2', s2)):
            if t is not None and (t.numel() != sum(pads) or t.dtype != f32 or not t.is_contiguous() or t.device != torch.device(device)):
                raise ValueError('%s does not have the [U | V | W | b] layout of this model (%d floats)' % (name, sum(pads)))
        self.flat_p = flat_p if flat_p is not None else torch.zeros(sum(pads), dtype=f32, device=device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.s1 = s1 if s1 is not None else (torch.zeros_like(self.flat_p) if self.opt_name != 'gd' else None)
        self.s2 = s2 if s2 is not None else (torch.zeros_like(self.flat_p) if self.opt_name == 'adam' else None)
        # Philox step word of rank r at optimizer step t (0-based) = word_base + t * G + r.  A caller that also draws from the
        # model's own call counter between steps (evaluation passes of the CLI) moves the base so that no word is used twice
        self.word_base = 0
        o, views, gviews, offs = 0, [], [], []
        for n, pd, shp in zip(sizes, pads, shapes):
            views.append(self.flat_p[o:o + n].view(shp))
            gviews.append(self.flat_g[o:o + n].view(shp))
            offs.append(o)
            o += pd
        self.U, self.V, self.W, self.b = views[:4]
        self.gU, self.gV, self.gW, self.gb = gviews[:4]
        self.extra = [(views[4 + 2 * k], views[5 + 2 * k]) for k in range(self.n_layers - 1)]
        self.gextra = [(gviews[4 + 2 * k], gviews[5 + 2 * k]) for k in range(self.n_layers - 1)]
        u8 = torch.uint8
        self.tU = torch.zeros((user_num + 3) // 4 * 4, dtype=u8, device=device)[:user_num]     # whole 32-bit words
        self.tV = torch.zeros((item_num + 3) // 4 * 4, dtype=u8, device=device)[:item_num]
        self.segments = [(offs[0], user_num, D, self.tU), (offs[1], item_num, D, self.tV)]
        self.dense_begin = offs[2]
        self.cap = max_rows * (S + 2)                       # users + true items + S candidates per row, at most
        self.words = self._buffer_words(self.cap, D, self.flat_p.numel() - self.dense_begin)
        self.buf = torch.zeros(self.words, dtype=f32, device=device)
        self.bufs = torch.zeros(world * self.words, dtype=f32, device=device)
        self.loss = torch.zeros(1, dtype=f32, device=device)
        self.loss_sum = torch.zeros(1, dtype=f32, device=device)
        self.t = 0
        # overlap mode: "touched by ANY rank" bytes + list (known before the step: schedule and candidate streams are
        # replicated), so that the optimizer pass over all the other rows can run while the exchange is in flight
        self.overlap = overlap
        self.gfU = torch.zeros((user_num + 3) // 4 * 4, dtype=u8, device=device)[:user_num]
        self.gfV = torch.zeros((item_num + 3) // 4 * 4, dtype=u8, device=device)[:item_num]
        self.gsegments = [(offs[0], user_num, D, self.gfU), (offs[1], item_num, D, self.gfV)]
        # prepared next step (HipBackend): second set of global bytes, this rank's de-duplication marks + row list
        self.gfU2, self.gfV2 = torch.zeros_like(self.gfU), torch.zeros_like(self.gfV)
        self.lfU, self.lfV = torch.zeros_like(self.gfU), torch.zeros_like(self.gfV)
        self.llist = torch.zeros(self.cap + 64, dtype=torch.int64, device=device)
        self.lcnt = torch.zeros(2, dtype=torch.int32, device=device)
        R = user_num + item_num                              # import tables of prepared steps, built one step ahead
        self.pmask = torch.zeros(2 * R, dtype=torch.int32, device=device)
        self.pwhere = torch.full((2 * world * R,), 0x7fffffff, dtype=torch.int32, device=device)
        self.next = None
        self.glist = torch.zeros(world * self.cap + 1024, dtype=torch.int64, device=device)
        self.gcnt = torch.zeros(2, dtype=torch.int32, device=device)
        self.parity = 0

    collectives = 'all_gather_into_tensor x1 per step, overlapped with the optimizer pass over the rows no rank touches'

    def crosscheck_replicas(self):
        """First contact with real RCCL at G > 1, outside any timed region (after the warm-up steps): the replicas must be
        bit-identical — every rank applied the same rank-ordered sums.  A checksum of the parameters is compared across the ranks;
        on a mismatch the reason goes to stderr, every rank takes rank 0's parameters and optimizer state and the step falls
        back to its synchronous form (no second stream beside the all-gather, no prepared next step): same results, the
        overlap is what is given up.  Returns True when the replicas agreed; self.collectives names what runs from here on."""
        import sys
        self.flush()
        chk = torch.stack([self.flat_p.double().sum(), self.flat_p.double().abs().sum()])
        lo, hi = chk.clone(), chk.clone()
        if self.G > 1:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if torch.equal(lo, hi):
            return True
        print('[dccf_amd.replicated] rank %d: replicas differ after the warm-up steps (checksum %r, min %r, max %r): resynchronising '
              'from rank 0 and falling back to the synchronous step' % (self.rank, chk.tolist(), lo.tolist(), hi.tolist()), file=sys.stderr)
        for t in (self.flat_p, self.s1, self.s2):
            if t is not None:
                dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        self.flat_g.zero_()
        for f in (self.tU, self.tV, self.gfU, self.gfV, self.gfU2, self.gfV2, self.lfU, self.lfV):
            f.zero_()
        self.overlap = False
        self.next = None
        self.collectives = 'all_gather_into_tensor x1 per step, synchronous (fallback: replicas differed after the warm-up with the overlapped form)'
        return False

    @staticmethod
    def _buffer_words(cap, D, nd):
        return (4 + 2 * cap + cap * D + nd + 3) // 4 * 4     # == dp_buffer_words (include/dccf_hip.h)

    def init_params(self, std=0.01):
        """BaseModel.init_paras (src/models/BaseModel.py:130-142): N(0, 0.01); the SAME values on every rank."""
        g = torch.Generator(device=self.dev).manual_seed(self.seed * 7919 + 7)
        for t in (self.U, self.V, self.W, self.b) + tuple(x for wb in self.extra for x in wb):
            t.normal_(0.0, std, generator=g)

    def set_params(self, U, V, W, b):
        for dst, src in ((self.U, U), (self.V, V), (self.W, W), (self.b, b)):
            dst.copy_(src)

    def train_step(self, X, Y, pred=None, X_all=None, X_all_next=None):
        """X int64 [2B, 2] = this rank's [positives ; negatives]; one optimizer step over the G ranks' batches.
        X_all int64 [G, 2B, 2] (every rank's batch of this step, X_all[rank] == X) enables the overlap of the optimizer
        pass over the rows NO rank touches with the all-gather.  X_all_next: the tensor the NEXT call will get as X_all
        (and X_all_next[rank] as X) — the optimizer launches then also prepare that step (same results, fewer launches).
        Returns (prediction of this rank's rows, loss summed over the ranks)."""
        be = self.be
        ov = self.overlap and X_all is not None
        self.next = None
        if ov and X_all_next is not None and tuple(X_all_next.shape) == tuple(X_all.shape):
            self.next = (X_all_next[self.rank], X_all_next)
        step0, t = self.word_base + self.t * self.G, self.t + 1
        pred = be.local(self, X, Y, step0 + self.rank, pred, X_all if ov else None, step0)
        work = None
        if self.G > 1 or _FORCE_COLLECTIVE:                        # the step's only collective
            work = dist.all_gather_into_tensor(self.bufs, self.buf, group=self.group, async_op=ov)
        else:
            self.bufs.copy_(self.buf)
        self.t = t
        if ov:
            be.overlap(self, t)
            if work is not None:
                work.wait()
        be.finish(self, t, ov)
        if ov:
            self.parity ^= 1
        return pred, self.loss_sum


def epoch_schedule(full, tail, G):
    """An epoch's batches (DeviceTrainSet.epoch_batches: full [nb, 2B, 2] = [positives ; negatives] per batch, tail [2r, 2] or
    None) as the steps of a G-rank job: (sched [ns, G, 2B, 2], last [G, 2b, 2] or None).  Step j < ns trains batches
    j G .. j G + G - 1, rank r the r-th of them.  What is left — fewer than G batches and the short last batch — becomes ONE more
    step of b = ceil(left / G) pairs per rank; the G b - left missing pairs (< G) are the epoch's first pairs again, as
    torch.utils.data.DistributedSampler completes an uneven epoch.  Every share keeps the batch layout: row k and row b + k are
    the positive and the negative of one pair (same uid, src/data_processor/DataProcessor.py:160-207)."""
    nb, B = full.shape[0], full.shape[1] // 2
    ns = nb // G
    sched = full[:ns * G].view(ns, G, 2 * B, 2)
    pos, neg = [full[ns * G:, :B].reshape(-1, 2)], [full[ns * G:, B:].reshape(-1, 2)]
    if tail is not None:
        r = tail.shape[0] // 2
        pos.append(tail[:r])
        neg.append(tail[r:])
    pos, neg = torch.cat(pos), torch.cat(neg)
    left = pos.shape[0]
    if left == 0:
        return sched, None
    b = (left + G - 1) // G
    pad = b * G - left
    if pad:
        src = full[0] if nb > 0 else torch.cat([pos, neg])
        h = src.shape[0] // 2
        take = torch.arange(pad, device=full.device) % h
        pos, neg = torch.cat([pos, src[:h][take]]), torch.cat([neg, src[h:][take]])
    return sched, torch.cat([pos.view(G, b, 2), neg.view(G, b, 2)], dim=1).contiguous()


# ------------------------------------------------------------------------------------------------------ bench entry
def bench_main(args, rank, world, dev):
    """bench.py --gpus N (N > 1): weak scaling — every rank trains `batch_size` pairs per step on a full replica."""
    import json
    import time
    from dccf_amd.data_processor import DeviceTrainSet
    from bench import synthetic_interactions
    from dccf_amd import utils
    U, I, D, F, B = args.users, args.items, args.dim, args.feat, args.batch_size
    S, A = 10, 2
    # everything up to the end of the warm-up (table setup, the first collectives on real RCCL) runs under a host-side deadline:
    # a hang at first contact makes every rank exit non-zero with a message instead of burning the launcher's time limit
    deadline = utils.Deadline(float(os.environ.get('DCCF_WARMUP_DEADLINE_S', '420')),
                              'set-up + %d warm-up steps of the replicated %d-rank step' % (args.warmup, world))
    be = HipBackend(dev)
    g = torch.Generator(device=dev).manual_seed(args.seed)          # same tables on every rank
    feat = torch.randn(I, F, generator=g, device=dev) * 0.05
    expo_mode = args.expo
    if expo_mode == 'auto':
        expo_mode = 'dense' if U * I * 4 < 160e9 else 'factors'
    expo, ips = None, None
    if expo_mode == 'dense':
        expo = torch.empty(U, I, device=dev)
        rows = max(1, (1 << 30) // (4 * I))
        for r0 in range(0, U, rows):
            expo[r0:r0 + rows].normal_(generator=g)
    else:
        ips = dict(P=torch.randn(U, 64, generator=g, device=dev) * 0.1, Q=torch.randn(I, 64, generator=g, device=dev) * 0.1,
                   bu=torch.randn(U, generator=g, device=dev) * 0.1, bi=torch.randn(I, generator=g, device=dev) * 0.1,
                   prop=torch.rand(I, generator=g, device=dev), b0=0.1, M=0.1)
    tr = ReplicatedDCCF(rank, world, U, I, D, S, A, 0.1, 0.2, 1e-3, 1e-4, args.seed, be, dev, feat, expo=expo, ips=ips,
                        max_rows=2 * B, overlap=bool(args.dp_overlap))
    tr.init_params()
    be.ctx.reserve(2 * B, D, F, S, A)
    n_steps = args.steps + args.warmup
    n_pairs = (n_steps + 2) * B * world
    uid, iid = synthetic_interactions(int(n_pairs * 1.15) + 1000, U, I, args.seed)      # replicated train set
    ds = DeviceTrainSet(uid[:n_pairs], iid[:n_pairs], U, I, args.seed)
    y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
    pred = torch.empty(2 * B, device=dev)

    def schedule(e, n):
        full, _ = ds.epoch_batches(e, B)                       # same permutation / negatives on every rank
        return full[:n * world].view(n, world, 2 * B, 2)       # step k: rank r trains full[k*world + r]

    pn = bool(getattr(args, 'prep_next', 1))

    def nxt(sched, k, n):                                      # the next step's schedule entry, when there is one
        return sched[k + 1] if pn and k + 1 < n else None

    sched = schedule(0, args.warmup)
    agreed_after_warmup = True
    for k in range(args.warmup):
        tr.train_step(sched[k, rank], y, pred, X_all=sched[k], X_all_next=nxt(sched, k, args.warmup) if k > 0 else None)
        if k == 0:
            # first contact, outside the timed region: after the FIRST warm-up step (the first collective on real RCCL) the replicas
            # must agree; if not -> stderr, resync from rank 0, synchronous step.  Placed here and not after the last warm-up step so
            # that the remaining warm-up steps run back to back into the timed region, as the contract's warm-up is meant to
            agreed_after_warmup = tr.crosscheck_replicas()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    deadline.cancel()
    deadline = utils.Deadline(float(os.environ.get('DCCF_BENCH_DEADLINE_S', '600')), 'the timed %d steps + the roofline section' % args.steps)
    t0 = time.perf_counter()
    sched = schedule(1, args.steps)                            # the epoch's negative sampling is timed
    for k in range(args.steps):
        tr.train_step(sched[k, rank], y, pred, X_all=sched[k], X_all_next=nxt(sched, k, args.steps))
    tr.flush()          # (timed: the rows the lazy regularisation left behind are brought up to date before the clock stops)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    tr.flush()
    # replicas must be bit-identical: compare a checksum of the parameters across ranks (outside the timed region)
    chk = torch.stack([tr.flat_p.double().sum(), tr.flat_p.double().abs().sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    # roofline: the kernels of the launch structure the timed region ran, bracketed by HIP events inside the library (dccf_profile)
    # over n_prof more steps of the same pipeline on every rank (the collective needs all of them), behind ~25 ms of queued work so
    # that no bracket contains a host launch gap; the cost of an event boundary (an empty bracket) is subtracted
    n_prof = max(1, min(args.steps + args.warmup, 50))
    sched = schedule(2, n_prof)
    blocker = torch.zeros(64 << 20, device=dev)
    for _ in range(64):
        blocker.add_(1.0)
    empties = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(n_prof)]
    be.ctx.profile(True)
    for k in range(n_prof):
        tr.train_step(sched[k, rank], y, pred, X_all=sched[k], X_all_next=nxt(sched, k, n_prof))
        empties[k][0].record()
        empties[k][1].record()
    tr.flush()
    torch.cuda.synchronize()
    prof = be.ctx.profile_read()
    be.ctx.profile(False)
    ev_ms = sum(a.elapsed_time(b) for a, b in empties) / n_prof
    k_ms = {k: max(v[0] / max(v[1], 1) - ev_ms, 1e-6) for k, v in prof.items()}
    L_rows = 2 * B * (S + 1) * A
    fwd_flops = 2.0 * L_rows * (D + F) * D
    bwd_flops = fwd_flops + 2.0 * L_rows * D * D
    MFMA_PEAK = 157.3                    # TFLOP/s, fp32 matrix (MI355X_MICROARCH.md), as in bench.py
    bwd_ms, fwd_ms = k_ms.get('noise_bwd_eps', 1e-6), k_ms.get('noise_fwd', 1e-6)
    # the dense regularised Adam pass over the whole replica, for the record (HBM-bound; p + m + v of this model fit the Infinity
    # Cache, so its figure is cache-assisted — bench.py's single-GPU line also measures it beyond the cache)
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(n_prof)]
    for k in range(n_prof):
        ev[k][0].record()
        be.L.dense_opt_phase(tr.opt_name, tr.flat_p, tr.flat_g, tr.s1, tr.s2, tr.lr, tr.l2, tr.l2, 50.0, tr.t + 1, tr.gsegments, 1)   # no row is marked: the whole pass
        ev[k][1].record()
        ev[k][2].record()          # empty bracket = the cost of an event boundary, contained once in the first bracket
    torch.cuda.synchronize()
    adam_ms = max(sum(a.elapsed_time(b) - b.elapsed_time(c) for a, b, c in ev) / n_prof, 1e-6)
    n_params = tr.flat_p.numel()
    roofline = {'kernel': 'k_bwd', 'bound': 'mfma', 'achieved': round(bwd_flops / 1e12 / (bwd_ms / 1e3), 2), 'peak': MFMA_PEAK,
                'unit': 'TFLOP/s', 'frac': round(bwd_flops / 1e12 / (bwd_ms / 1e3) / MFMA_PEAK, 4), 'traffic': None,
                'algorithmic_per_launch': round(bwd_flops / 1e12, 6), 'avg_launch_ms': round(bwd_ms, 5),
                'launch_structure': 'the replicated step (dccf_dp_local | all-gather || dccf_dp_overlap | dccf_dp_finish): forward and '
                                    'backward bracketed by HIP events inside the library on the launch stream on rank 0; event '
                                    'boundary %.4f ms subtracted' % ev_ms,
                'noise_fwd': {'avg_launch_ms': round(fwd_ms, 5), 'TFLOPs': round(fwd_flops / 1e12 / (fwd_ms / 1e3), 2),
                              'frac': round(fwd_flops / 1e12 / (fwd_ms / 1e3) / MFMA_PEAK, 4)},
                'dense_adam_whole_pass': {'bound': 'hbm', 'achieved': round(24.0 * n_params / 1e9 / (adam_ms / 1e3), 2), 'peak': 8000.0,
                                          'unit': 'GB/s', 'frac': round(24.0 * n_params / 1e9 / (adam_ms / 1e3) / 8000.0, 4),
                                          'algorithmic_GB': round(24.0 * n_params / 1e9, 4), 'avg_launch_ms': round(adam_ms, 5),
                                          'note': 'cache-assisted (p + m + v fit the Infinity Cache); not a launch of the default '
                                                  '(lazy) step'}}
    # the step's collective ALONE (diagnostic, outside the timed region, every rank takes part): n_ag all-gathers of the exchange buffer
    # back to back on an idle GPU — what the step has to hide behind its first optimizer phase (a device copy at world size 1)
    n_ag = 20
    dist.barrier()
    torch.cuda.synchronize()
    ag0, ag1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ag0.record()
    for _ in range(n_ag):
        if world > 1 or _FORCE_COLLECTIVE:
            dist.all_gather_into_tensor(tr.bufs, tr.buf, group=tr.group)
        else:
            tr.bufs.copy_(tr.buf)
    ag1.record()
    torch.cuda.synchronize()
    ag_us = ag0.elapsed_time(ag1) / n_ag * 1e3
    if rank == 0:
        out = {'metric': 'train pairs/sec at rank=64 Electronics', 'value': round(args.steps * B * world / dt, 1),
               'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'DCCF train step (fwd + BPR + bwd + dense l2/clip/Adam), Electronics-shaped synthetic: '
                                      'user_num=%d item_num=%d D=%d F=%d S=%d A=%d, exposure=%s, layout=replicated (full replica per '
                                      'GPU, one all-gather of the touched gradient rows per step; --mp sharded runs the row-sharded '
                                      'all-to-all layout), fused on-device negatives' % (U, I, D, F, S, A, expo_mode),
                          'layout': 'replicated',
                          'regularisation': ('windowed lazy (K = %d)' % be.lazy.K) if getattr(be, 'lazy', None) is not None else 'dense pass',
                          'batch_size_per_gpu': B, 'global_batch': B * world, 'optimizer': 'Adam lr=1e-3 l2=1e-4 dropout=0.2',
                          'collectives_per_step': 'all_gather x1 (touched gradient rows + [dW|db], %.2f MB per rank)'
                                                  % (tr.words * 4 / 1e6),
                          'collectives': tr.collectives, 'all_gather_alone_us': round(ag_us, 2),
                          'step_kernels_us': {'noise_fwd': round(fwd_ms * 1e3, 2), 'k_bwd': round(bwd_ms * 1e3, 2)},
                          'replicas_agreed_after_warmup': bool(agreed_after_warmup),
                          'replicas_bit_identical': bool(torch.equal(lo, hi))},
               'roofline': roofline, 'cpu_baseline': None}
        import bench
        bench.emit(out)
    deadline.cancel()
    dist.destroy_process_group()
