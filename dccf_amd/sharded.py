# coding=utf-8
"""Row-sharded DCCF training across the GPUs of one node (BASELINE.json config 5).

The reference is single-GPU (src/main.py:106,153-155); this is new capability with no reference counterpart.

Partitioning.  One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Pairs are data-parallel:
each rank trains ``batch_size`` pairs per step.  Embedding rows are sharded cyclically, owner(row) = row mod G, local
index = row div G; the frozen per-item tables (768-d features, IPS factors that give the exposure score on the fly —
a dense U x I matrix is impossible at 10M x 1M) live with the item shard, the per-user ones with the user shard.  The
dense ``W, b`` are replicated.  Optimizer state lives with the shard, so the dense regularised Adam pass is local.

Exchange per step (the path's only real exchange steps):
  1. all-to-all (rows): every rank receives the user / candidate-item / feature rows its pairs touch;
  2. all-to-all (grad rows): the per-slot gradient rows travel back and are summed into the owner's gradient shard;
  3. all-reduce of ``dW, db`` (213 KB at D=64).
There is NO id exchange: the train set is replicated (16 B per interaction), negatives and the shuffle are
deterministic functions of (seed, epoch), candidates are a counter-based Philox stream of (seed, step, global row) —
so every rank computes every rank's batch and knows which of its rows each peer needs, in a canonical slot order.

The local compute is the single-GPU HIP path unchanged: the received rows form compact per-step tables (compact id =
slot index) and ``dccf_train_fwdbwd`` runs on them (rnd.mode 2: injected candidates, fused noise / dropout).
"""
import torch
import torch.distributed as dist


class HipBackend(object):
    """The product backend: libdccf_hip.so through dccf_amd._lib."""

    def __init__(self, device):
        from dccf_amd import _lib
        self.L = _lib
        self.device = device
        self.ctx = _lib.Context(device.index or 0)

    def candidates(self, n_rows, S, item_num, seed, step):
        return self.L.debug_candidates(n_rows, S, item_num, seed, step, self.device)

    def local_step(self, Uc, Vc, W, b, featc, ips, Xc, cand_c, Y, S, A, std, dropout, seed, step, gU, gV, gW, gb):
        m = self.L.model_struct(Uc, Vc, W, b, featc, None, S, A, std, ips=ips)
        r = self.L.rand_struct(sample_item=cand_c, seed=seed, step=step)
        return self.L.dccf_train_fwdbwd(self.ctx, m, r, Xc, Y, 1, dropout, gU, gV, gW, gb)

    def opt_step(self, p, g, s1, s2, lr, l2, t):
        self.L.dense_opt_step('adam', p, g, s1, s2, lr, l2, l2, 50.0, t, zero_grad=True)


def _a2a(out, inp, out_splits, in_splits, group):
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


class ShardedDCCF(object):
    def __init__(self, rank, world, user_num, item_num, D, S, A, std, dropout, lr, l2, seed, backend, device,
                 feat_local, ips_local, group=None):
        """feat_local: [ceil(item_num/G), F] rows of the items i = rank (mod G); ips_local: dict P [nU_loc,Dq], bu [nU_loc],
        Q [nI_loc,Dq], bi [nI_loc], prop [nI_loc], b0, M — the IPSBiasedMF factors of the exposure score."""
        self.rank, self.G, self.group, self.dev, self.be = rank, world, group, device, backend
        self.user_num, self.item_num, self.D, self.S, self.A = user_num, item_num, D, S, A
        self.std, self.dropout, self.lr, self.l2, self.seed = std, dropout, lr, l2, seed
        self.F = feat_local.shape[1]
        self.Dq = ips_local['P'].shape[1]
        self.nU = (user_num + world - 1 - rank) // world if user_num > rank else 0
        self.nI = (item_num + world - 1 - rank) // world if item_num > rank else 0
        self.feat, self.ips = feat_local, ips_local
        sizes = [self.nU * D, self.nI * D, D * (D + self.F), D]
        pads = [(n + 3) // 4 * 4 for n in sizes]
        f32 = torch.float32
        self.flat_p = torch.zeros(sum(pads), dtype=f32, device=device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.s1 = torch.zeros_like(self.flat_p)
        self.s2 = torch.zeros_like(self.flat_p)
        o, views, gviews = 0, [], []
        for n, pd, shp in zip(sizes, pads, [(self.nU, D), (self.nI, D), (D, D + self.F), (D,)]):
            views.append(self.flat_p[o:o + n].view(shp))
            gviews.append(self.flat_g[o:o + n].view(shp))
            o += pd
        self.U, self.V, self.W, self.b = views
        self.gU, self.gV, self.gW, self.gb = gviews
        self.t = 0
        # user-side payload row: [U | P | bu], item-side: [V | Q | bi | prop]
        self.wu, self.wi = D + self.Dq + 1, D + self.Dq + 2

    def init_params(self, std=0.01):
        """BaseModel.init_paras (src/models/BaseModel.py:130-142): N(0, 0.01); W, b identical on every rank."""
        g = torch.Generator(device=self.dev).manual_seed(self.seed * 7919 + 13 + self.rank)
        self.U.normal_(0.0, std, generator=g)
        self.V.normal_(0.0, std, generator=g)
        g2 = torch.Generator(device=self.dev).manual_seed(self.seed * 7919 + 7)
        self.W.normal_(0.0, std, generator=g2)
        self.b.normal_(0.0, std, generator=g2)

    def set_global_params(self, U, V, W, b):
        """Takes the FULL tables (tests): keeps this rank's rows."""
        self.U.copy_(U[self.rank::self.G])
        self.V.copy_(V[self.rank::self.G])
        self.W.copy_(W)
        self.b.copy_(b)

    # ------------------------------------------------------------------------------------------------ one step
    def _route(self, ids):
        """ids: int64 [G, T] — the slots of every rank (same tensor on every rank).  Returns what this rank sends
        (local row indices, ordered by destination then slot) and how what it receives maps to its own slots."""
        me = self.rank
        owner = ids % self.G
        mine = owner == me                                   # [G, T]: slots of rank q whose rows I own
        send_counts = mine.sum(1)
        send_lidx = (ids // self.G)[mine]                    # row-major: destination q, then slot order
        my_owner = owner[me]
        recv_perm = torch.argsort(my_owner, stable=True)     # received row j belongs to my slot recv_perm[j]
        recv_counts = torch.bincount(my_owner, minlength=self.G)
        return send_lidx, send_counts, recv_perm, recv_counts

    def _fetch(self, tables, route, width):
        """tables: list of [rows] or [rows, w] tensors sharing the row index; their selected rows travel side by side."""
        send_lidx, sc, perm, rc = route
        send = torch.cat([t.index_select(0, send_lidx).view(send_lidx.numel(), -1) for t in tables], 1)
        recv = torch.empty((int(perm.numel()), width), dtype=send.dtype, device=self.dev)
        _a2a(recv, send, rc, sc, self.group)
        out = torch.empty_like(recv)
        out[perm] = recv
        return out

    def _push(self, grad_rows, route, gtable):
        send_lidx, sc, perm, rc = route
        send = grad_rows.index_select(0, perm)               # back in (owner, slot) order
        recv = torch.empty((int(send_lidx.numel()), grad_rows.shape[1]), dtype=grad_rows.dtype, device=self.dev)
        _a2a(recv, send, sc, rc, self.group)
        gtable.index_add_(0, send_lidx, recv)

    def train_step(self, X_all, step):
        """X_all: int64 [G, 2B, 2] global ids — every rank's [pos ; neg] batch (identical on all ranks).
        Returns (prediction [2B], loss [1]) of THIS rank's pairs."""
        G, D, S, A = self.G, self.D, self.S, self.A
        S1 = S + 1
        N = X_all.shape[1]
        B = N // 2
        cand_all = self.be.candidates(G * N, S, self.item_num, self.seed, step).view(G, N, S)
        users = X_all[:, :B, 0]                                                    # [G, B]   (rows k and B+k share it)
        items = torch.cat([X_all[:, :, 1:2], cand_all], 2).reshape(G, N * S1)        # [G, N*S1] candidate slots
        feats = X_all[:, :, 1]                                                     # [G, N]   true items
        ru, ri, rf = self._route(users), self._route(items), self._route(feats)
        splits = torch.stack([ru[1], ru[3], ri[1], ri[3], rf[1], rf[3]]).tolist()  # one host sync per step
        ru = (ru[0], splits[0], ru[2], splits[1])
        ri = (ri[0], splits[2], ri[2], splits[3])
        rf = (rf[0], splits[4], rf[2], splits[5])
        ips = self.ips
        urows = self._fetch([self.U, ips['P'], ips['bu']], ru, self.wu)
        irows = self._fetch([self.V, ips['Q'], ips['bi'], ips['prop']], ri, self.wi)
        frows = self._fetch([self.feat], rf, self.F)
        Dq = self.Dq
        Uc, Vc = urows[:, :D].contiguous(), irows[:, :D].contiguous()
        ipsc = dict(P=urows[:, D:D + Dq].contiguous(), bu=urows[:, D + Dq].contiguous(),
                    Q=irows[:, D:D + Dq].contiguous(), bi=irows[:, D + Dq].contiguous(),
                    prop=irows[:, D + Dq + 1].contiguous(), b0=ips['b0'], M=ips['M'])
        featc = torch.zeros((N * S1, self.F), dtype=torch.float32, device=self.dev)
        featc[0::S1] = frows                                                        # true item of row n has compact id n*S1
        ar = torch.arange(N, device=self.dev)
        Xc = torch.stack([ar % B, ar * S1], 1).contiguous()
        cand_c = (ar.view(N, 1) * S1 + torch.arange(1, S1, device=self.dev).view(1, S)).contiguous()
        Y = torch.cat([torch.ones(B, device=self.dev), torch.zeros(B, device=self.dev)])
        gUc, gVc = torch.zeros_like(Uc), torch.zeros_like(Vc)
        pred, loss = self.be.local_step(Uc, Vc, self.W, self.b, featc, ipsc, Xc, cand_c, Y, S, A, self.std, self.dropout,
                                        self.seed, step * G + self.rank, gUc, gVc, self.gW, self.gb)
        self._push(gUc, ru, self.gU)
        self._push(gVc, ri, self.gV)
        dense = torch.cat([self.gW.reshape(-1), self.gb])
        dist.all_reduce(dense, group=self.group)
        self.gW.copy_(dense[:self.gW.numel()].view_as(self.gW))
        self.gb.copy_(dense[self.gW.numel():])
        self.t += 1
        self.be.opt_step(self.flat_p, self.flat_g, self.s1, self.s2, self.lr, self.l2, self.t)
        return pred, loss


# ------------------------------------------------------------------------------------------------------ bench entry
def bench_main(args, rank, world, dev):
    """bench.py --gpus N (N > 1): weak scaling — every rank trains `batch_size` pairs per step on its shard."""
    import json
    import time
    import numpy as np
    from dccf_amd.data_processor import DeviceTrainSet
    U, I, D, F, B = args.users, args.items, args.dim, args.feat, args.batch_size
    S, A = 10, 2
    be = HipBackend(dev)
    g = torch.Generator(device=dev).manual_seed(args.seed + 1000 * rank)
    nU, nI = (U + world - 1 - rank) // world, (I + world - 1 - rank) // world
    feat = torch.randn(nI, F, generator=g, device=dev) * 0.05
    ips = dict(P=torch.randn(nU, 64, generator=g, device=dev) * 0.1, Q=torch.randn(nI, 64, generator=g, device=dev) * 0.1,
               bu=torch.randn(nU, generator=g, device=dev) * 0.1, bi=torch.randn(nI, generator=g, device=dev) * 0.1,
               prop=torch.rand(nI, generator=g, device=dev), b0=0.1, M=0.1)
    tr = ShardedDCCF(rank, world, U, I, D, S, A, 0.1, 0.2, 1e-3, 1e-4, args.seed, be, dev, feat, ips)
    tr.init_params()
    from bench import synthetic_interactions
    n_pairs = (args.steps + args.warmup + 2) * B * world
    uid, iid = synthetic_interactions(int(n_pairs * 1.15) + 1000, U, I, args.seed)      # replicated train set
    ds = DeviceTrainSet(uid[:n_pairs], iid[:n_pairs], U, I, args.seed)

    def epoch(e):
        full, _ = ds.epoch_batches(e, B)                       # same permutation / negatives on every rank
        nb = full.shape[0] // world * world
        return full[:nb].view(nb // world, world, 2 * B, 2)     # step k: rank r trains full[k*world + r]

    sched = epoch(0)
    for k in range(args.warmup):
        tr.train_step(sched[k], k)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sched = epoch(1)
    for k in range(args.warmup, args.warmup + args.steps):
        tr.train_step(sched[k], k)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    if rank == 0:
        out = {'metric': 'train pairs/sec at rank=64 Electronics', 'value': round(args.steps * B * world / dt, 1),
               'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'DCCF train step, Electronics-shaped synthetic: user_num=%d item_num=%d D=%d F=%d S=%d '
                                      'A=%d, rows sharded mod %d, exposure from IPS factors, fused on-device negatives'
                                      % (U, I, D, F, S, A, world),
                          'batch_size_per_gpu': B, 'global_batch': B * world, 'optimizer': 'Adam lr=1e-3 l2=1e-4 dropout=0.2',
                          'collectives_per_step': 'all_to_all x5 (rows, grad rows) + all_reduce(dW,db)'},
               'roofline': None, 'cpu_baseline': None}
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()
