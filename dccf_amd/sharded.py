# coding=utf-8
"""Row-sharded DCCF training across the GPUs of one node (BASELINE.json config 5).

The reference is single-GPU (src/main.py:106,153-155); this is new capability with no reference counterpart.

Partitioning.  One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Pairs are data-parallel:
each rank trains ``batch_size`` pairs per step.  Embedding rows are sharded cyclically, owner(row) = row mod G, local
index = row div G; the frozen per-item tables (768-d features, IPS factors that give the exposure score on the fly —
a dense U x I matrix is impossible at 10M x 1M) live with the item shard, the per-user ones with the user shard.  The
dense ``W, b`` are replicated.  Optimizer state lives with the shard, so the dense regularised Adam pass is local.

Exposure (src/models/DCCF.py:98).  Two forms, the same numbers: (a) IPSBiasedMF factors sharded with the tables — a dense
U x I matrix is impossible at 10M x 1M — from which Expo[u, i] is computed on the fly; (b) the dense `<ds>.ips_expo_prob.npy`
of the reference sharded BY USER ROWS (rank r holds rows u = r mod G: 1/G of the 48.5 GB at Electronics size): the owner of a
user row gathers the 2 (S + 1) exposures of the pair that user belongs to — it knows the pair's items and candidates, the
schedule is replicated — and ships them behind the embedding row (`dccf_model_t.expo_gathered`).

Exchange per step — the path's only real exchange steps, 4 collectives:
  * all-to-all (rows): the user and candidate-item rows this rank's pairs touch, as ONE payload (both kinds are
    ``[embedding | IPS factor | bias | propensity]`` rows), and a second all-to-all for the 768-d true-item feature rows;
  * all-to-all (gradient rows): per-slot gradient rows back to their owners, summed there with float atomics;
  * all-reduce of the contiguous ``[dW | db]`` slice of the flat gradient buffer (213 KB at D=64: latency-bound).
There is NO id exchange: the train set is replicated (16 B per interaction), negatives and the shuffle are
deterministic functions of (seed, epoch), candidates are a counter-based Philox stream of (seed, epoch, global row) —
so every rank computes every rank's batches and knows which of its rows each peer needs, in a canonical slot order.
The routing tables of ALL steps of an epoch are computed once, vectorised (``EpochPlan``); a step then only launches
``shard_pack_rows`` -> all-to-all -> ``shard_unpack_rows`` -> the unchanged single-GPU kernels on compact per-step tables
(compact id = position in the receive buffer, so nothing is permuted) -> all-to-all -> ``shard_scatter_add`` -> Adam.
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

    # Row movers: several pack (unpack) jobs are ONE launch.  `make_jobs` builds the argument block once per epoch plan from
    # [(idx, dst, n, tables, payload)]; a step only rewrites (idx, dst, n) of each job.
    def make_jobs(self, jobs):
        return self.L.shard_jobs(jobs)

    def set_job(self, jobs, q, idx, dst, n):
        jobs[q].idx, jobs[q].dst, jobs[q].n = self.L.ptr(idx, torch.int32), self.L.ptr(dst, torch.int32), int(n)

    def pack_multi(self, jobs):
        self.L.shard_pack_multi(jobs)

    def unpack_multi(self, jobs, zero=None):
        self.L.shard_unpack_multi(jobs, zero)

    def scatter_add(self, idx, n, rows, g, flags=None):
        self.L.shard_scatter_add(idx, n, rows, g, flags)

    def local_step(self, Uc, Vc, W, b, featc, ips, Xc, cand_c, Y, S, A, std, dropout, seed, step, gU, gV, gW, gb, pred=None,
                   loss=None, extra=None, gextra=None, eg=None):
        """extra / gextra: the Linear(D, D) layers of --n_layers > 1 and their gradients; eg [N, S + 1]: the exposures of this
        batch's (row, candidate) slots as the owners of the user rows gathered them (dense exposure sharded by user rows)."""
        # the compact tables of an epoch plan keep their addresses: the argument blocks are built once per plan
        key = (Uc.data_ptr(), featc.data_ptr(), W.data_ptr(), gU.data_ptr(), gW.data_ptr(), 0 if eg is None else eg.data_ptr())
        if getattr(self, '_key', None) != key:
            self._m = self.L.model_struct(Uc, Vc, W, b, featc, None, S, A, std, ips=ips, extra=extra, expo_gathered=eg)
            self._r = self.L.rand_struct(sample_item=cand_c, seed=seed, step=step)
            self._key = key
        self._r.sample_item, self._r.step = self.L.ptr(cand_c, torch.int64), int(step)
        return self.L.dccf_train_fwdbwd(self.ctx, self._m, self._r, Xc, Y, 1, dropout, gU, gV, gW, gb, pred=pred, loss=loss,
                                        gextra=gextra)

    # windowed lazy regularisation of the shard (include/dccf_hip.h, dccf_opt_t.lazy_*): the rows this rank is about to send are
    # brought up to date before they are packed; the optimizer launch then takes them (with the gradient rows that came back),
    # W, b and one K-th of the shard's other rows — at BASELINE config 5's size the dense pass over the shard IS the step
    def lazy_state(self, p, g, s1, s2, lr, l2, segments, K, list_cap, kind='adam'):
        opt = self.L.opt_struct(kind, p, g, s1, s2, lr, l2, l2, 50.0, segments, 0)
        n_rows = sum(int(s[1]) for s in segments)
        st = self.L.LazyState(opt, K, n_rows, list_cap, lr, p.device)
        st._opt_keep = opt
        return st

    def opt_step(self, p, g, s1, s2, lr, l2, t, segments=None, kind='adam'):
        """torch.optim.{SGD, Adagrad, Adam}(lr, weight_decay=l2) + the explicit l2 term + clip (src/runners/BaseRunner.py:83-107,
        181-187) over this rank's shard."""
        if segments:       # rows no peer sent a gradient for: g neither read nor re-zeroed (24 instead of 32 B/param)
            self.L.dense_opt_step_rows(kind, p, g, s1, s2, lr, l2, l2, 50.0, t, segments)
        else:
            self.L.dense_opt_step(kind, p, g, s1, s2, lr, l2, l2, 50.0, t, zero_grad=True)


def _first_n(mask, values, nmax):
    """Per row of `mask` [nb, M]: the `values` of the True entries in order, left-aligned and padded to nmax."""
    order = torch.argsort((~mask).to(torch.int8), dim=1, stable=True)[:, :nmax]
    return [torch.gather(v, 1, order).to(torch.int32).contiguous() for v in values]


class _Route(object):
    """Routing of one payload kind for every step of an epoch.  A kind is a list of T slots per rank whose first
    `n_a` slots index table A (users) and the rest table B (items) — or a single table when n_a == T."""

    def __init__(self, ids, n_a, row_off_b, me, G):
        nb, _, T = ids.shape                                   # ids: int64 [nb, G, T], identical on every rank
        dev = ids.device
        owner, lidx = ids % G, ids // G
        is_a = (torch.arange(T, device=dev) < n_a).expand(nb, G, T)
        mine = owner == me                                     # slots of rank q (row-major q, t) whose rows I own
        mf = mine.reshape(nb, G * T)
        pos = (torch.cumsum(mf, 1) - 1)                        # position of a sent row in my send buffer: (q, t) order
        lf, af = lidx.reshape(nb, G * T), is_a.reshape(nb, G * T)
        counts = mine.sum(2)                                   # [nb, G] rows I send to q
        self.send_splits, self.send_n = counts.tolist(), counts.sum(1).tolist()
        self.send_np = counts.to(torch.int64).cpu().numpy().copy()             # [nb, G] rows per peer (the direct RCCL path)
        self.send_ad = self.send_np.ctypes.data                                # (row k of it: + 8 G k)
        self.send_max = max(1, max(self.send_n)) if nb else 1
        na, nbb = (mf & af).sum(1), (mf & ~af).sum(1)
        self.na, self.nb_ = na.tolist(), nbb.tolist()
        flat = torch.arange(G * T, device=dev).expand(nb, G * T)
        # a_slot: which (rank q, slot t) = q T + t of the schedule a sent table-A row serves (the owner-side exposure gather)
        self.a_src, self.a_dst, self.a_slot = _first_n(mf & af, [lf, pos, flat], max(1, int(na.max()) if nb else 1))
        self.b_src, self.b_dst = _first_n(mf & ~af, [lf, pos], max(1, int(nbb.max()) if nb else 1))
        # where a returned gradient row (send order) lands in the flat gradient buffer viewed as rows of width D
        (self.g_row,) = _first_n(mf, [torch.where(af, lf, lf + row_off_b)], self.send_max)
        my_owner = owner[:, me, :]                             # [nb, T]
        perm = torch.argsort(my_owner, dim=1, stable=True)     # receive position j holds my slot perm[j]
        self.perm = perm
        self.inv = torch.empty_like(perm)
        self.inv.scatter_(1, perm, torch.arange(T, device=dev).expand(nb, T))    # slot t sits at inv[t]
        rs = torch.stack([(my_owner == o).sum(1) for o in range(G)], 1)
        self.recv_splits = rs.tolist()
        self.recv_np = rs.to(torch.int64).cpu().numpy().copy()
        self.recv_ad = self.recv_np.ctypes.data
        self.T = T


def _lcm(a, b):
    import math
    return a * b // math.gcd(a, b)


class ShardedDCCF(object):
    def __init__(self, rank, world, user_num, item_num, D, S, A, std, dropout, lr, l2, seed, backend, device,
                 feat_local, ips_local=None, group=None, lazy_K=None, direct=None, expo_local=None, opt_name='adam', n_layers=1):
        """feat_local: [ceil(item_num/G), F] rows of the items i = rank (mod G).  Exposure, one of: ips_local — dict P [nU_loc,Dq],
        bu [nU_loc], Q [nI_loc,Dq], bi [nI_loc], prop [nI_loc], b0, M: the IPSBiasedMF factors of the exposure score, sharded
        like the tables; expo_local — [nU_loc, item_num]: the rows u = rank (mod G) of the dense exposure matrix.
        opt_name: 'adam' | 'adagrad' | 'gd' (src/runners/BaseRunner.py:83-107).  n_layers: the extra Linear(D, D) layers of
        --n_layers > 1 (src/models/DMF.py:14, src/models/DCCF.py:61-62) are replicated like W, b and follow them in the flat
        buffer, weight then bias per layer: their gradients ride in the all-reduce of the dense tail."""
        self.rank, self.G, self.group, self.dev, self.be = rank, world, group, device, backend
        self.user_num, self.item_num, self.D, self.S, self.A = user_num, item_num, D, S, A
        self.std, self.dropout, self.lr, self.l2, self.seed = std, dropout, lr, l2, seed
        self.opt_name, self.n_layers = opt_name.lower(), int(n_layers)
        if self.opt_name not in ('adam', 'adagrad', 'gd'):
            raise ValueError('unknown optimizer ' + opt_name)
        if (ips_local is None) == (expo_local is None):
            raise ValueError('exposure: give the sharded IPS factors (ips_local) or the local rows of the dense matrix (expo_local)')
        if not 1 <= D <= 256 or not 1 <= self.n_layers <= 8 or (D > 128 and self.n_layers > 1):
            raise ValueError('embedding size 1 .. 256 (above 128: n_layers 1), n_layers 1 .. 8')
        self.F = feat_local.shape[1]
        self.nU = (user_num + world - 1 - rank) // world if user_num > rank else 0
        self.nI = (item_num + world - 1 - rank) // world if item_num > rank else 0
        self.feat, self.ips, self.expo = feat_local.contiguous(), ips_local, expo_local
        self.dense_expo = expo_local is not None
        self.Dq = 0 if self.dense_expo else ips_local['P'].shape[1]
        if self.dense_expo and tuple(expo_local.shape) != (max(self.nU, 0), item_num):
            raise ValueError('expo_local must be [rows u = rank (mod G), item_num]')
        sizes = [self.nU * D, self.nI * D, D * (D + self.F), D] + [D * D, D] * (self.n_layers - 1)
        shapes = [(self.nU, D), (self.nI, D), (D, D + self.F), (D,)] + [(D, D), (D,)] * (self.n_layers - 1)
        unit = _lcm(256, D)                                       # tables start on 256-float bounds AND on whole rows of [dU ; dV]
        pads = [(n + unit - 1) // unit * unit for n in sizes[:2]] + [(n + 255) // 256 * 256 for n in sizes[2:]]
        f32 = torch.float32
        self.flat_p = torch.zeros(sum(pads), dtype=f32, device=device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.s1 = torch.zeros_like(self.flat_p) if self.opt_name != 'gd' else None      # Adam's m / Adagrad's sum
        self.s2 = torch.zeros_like(self.flat_p) if self.opt_name == 'adam' else None     # Adam's v
        o, views, gviews, self.offs = 0, [], [], []
        for n, pd, shp in zip(sizes, pads, shapes):
            views.append(self.flat_p[o:o + n].view(shp))
            gviews.append(self.flat_g[o:o + n].view(shp))
            self.offs.append(o)
            o += pd
        self.sizes, self.pads = sizes, pads
        self.U, self.V, self.W, self.b = views[:4]
        self.gU, self.gV, self.gW, self.gb = gviews[:4]
        self.extra = [(views[4 + 2 * k], views[5 + 2 * k]) for k in range(self.n_layers - 1)]
        self.gextra = [(gviews[4 + 2 * k], gviews[5 + 2 * k]) for k in range(self.n_layers - 1)]
        self.g_rows = self.flat_g[:pads[0] + pads[1]].view(-1, D)          # [dU ; dV] shards as rows of width D
        self.row_off_v = pads[0] // D
        # one "touched" byte per row of [dU ; dV] (set by the scatter-add of the received gradient rows) for the row-aware
        # optimizer step; the two shards are ONE segment (the padding rows between them are never flagged)
        n_rows = (pads[0] + pads[1]) // D
        self.touched = torch.zeros((n_rows + 3) // 4 * 4, dtype=torch.uint8, device=device)[:n_rows]
        self.segments = [(0, n_rows, D, self.touched)] if (D % 4 == 0 and D >= 4) else None      # (rows of whole float4 slots)
        self.dense_begin = pads[0] + pads[1]
        self.g_dense = self.flat_g[self.dense_begin:]                       # [dW | db | dW_1 | db_1 ...] (+ zero padding), contiguous
        self.user_pad = torch.ones((max(self.nU, 1), 1), dtype=f32, device=device)           # the "prop" column of user rows
        self.t = 0
        self.word_base = 0               # Philox step word of rank r at optimizer step t (0-based) = word_base + t G + r
        self.plan = None
        import os
        self.lazy_K = int(os.environ.get('DCCF_LAZY_K', '8')) if lazy_K is None else int(lazy_K)
        self.lazy = None                 # created with the first epoch plan (the row list is sized by it)
        # collectives straight on RCCL (one C call each on the launch stream) when the process group is RCCL; through
        # torch.distributed otherwise (gloo: the CPU tests and the one-GPU rehearsals) or with DCCF_SHARD_DIRECT=0
        if direct is None:
            direct = os.environ.get('DCCF_SHARD_DIRECT', '1') != '0'
        self.comm = None
        self.collectives = 'torch.distributed'
        if direct and hasattr(backend, 'L') and dist.is_initialized() and dist.get_backend(group) == 'nccl':
            try:
                self.comm = backend.L.Comm(rank, world, device, group)
                self.collectives = 'rccl-direct'
            except RuntimeError as e:        # (no RCCL copy mapped, communicator creation failed): the c10d branch does the same work
                import sys
                print('[dccf_amd.sharded] direct RCCL communicator unavailable (%s): collectives go through torch.distributed' % e,
                      file=sys.stderr)
                self.collectives = 'torch.distributed (no direct communicator: %s)' % e

    def views_of(self, flat):
        """[U shard, V shard, W, b, W_1, b_1, ...] as views of a flat tensor laid out like flat_p (the optimizer state buffers)."""
        shapes = [(self.nU, self.D), (self.nI, self.D), (self.D, self.D + self.F), (self.D,)] + [(self.D, self.D), (self.D,)] * (self.n_layers - 1)
        return [flat[o:o + n].view(shp) for o, n, shp in zip(self.offs, self.sizes, shapes)]

    def close(self):
        """Destroys the directly created RCCL communicator (before torch.distributed's process group goes away)."""
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def flush(self):
        """Every row of the shard up to date (before anything but train_step reads U, V or the optimizer state)."""
        if self.lazy is not None:
            self.lazy.flush(self.t)

    def init_params(self, std=0.01):
        """BaseModel.init_paras (src/models/BaseModel.py:130-142): N(0, 0.01); W, b identical on every rank."""
        g = torch.Generator(device=self.dev).manual_seed(self.seed * 7919 + 13 + self.rank)
        self.U.normal_(0.0, std, generator=g)
        self.V.normal_(0.0, std, generator=g)
        g2 = torch.Generator(device=self.dev).manual_seed(self.seed * 7919 + 7)
        for t in (self.W, self.b) + tuple(x for wb in self.extra for x in wb):
            t.normal_(0.0, std, generator=g2)

    def set_global_params(self, U, V, W, b, extra=None):
        """Takes the FULL tables (tests, the CLI): keeps this rank's rows."""
        self.flush()
        self.U.copy_(U[self.rank::self.G])
        self.V.copy_(V[self.rank::self.G])
        self.W.copy_(W)
        self.b.copy_(b)
        for (w, bb), (w0, b0) in zip(self.extra, extra or []):
            w.copy_(w0)
            bb.copy_(b0)

    def gather_tables(self, U_out, V_out):
        """Writes the full [user_num, D] / [item_num, D] tables (this rank's rows + every peer's: one all-gather per table) —
        what a caller needs to evaluate or checkpoint with single-GPU code.  Collective."""
        self.flush()
        for mine, out, n in ((self.U, U_out, self.user_num), (self.V, V_out, self.item_num)):
            nmax = (n + self.G - 1) // self.G
            if self.G == 1:
                out.copy_(mine)
                continue
            buf = torch.zeros((nmax, self.D), dtype=mine.dtype, device=mine.device)
            buf[:mine.shape[0]] = mine
            allb = torch.empty((self.G, nmax, self.D), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(allb.view(-1), buf.view(-1), group=self.group)
            for r in range(self.G):
                out[r::self.G] = allb[r, :len(range(r, n, self.G))]

    # ------------------------------------------------------------------------------------------------ epoch plan
    def begin_epoch(self, X_sched, epoch):
        """X_sched: int64 [n_steps, G, 2B, 2] global ids — step k, rank q trains X_sched[k, q] = [pos ; neg] (the same
        tensor on every rank); `epoch`: the Philox step word of the epoch's candidate stream.  Builds the routing tables of
        every step of the epoch in one vectorised pass."""
        nb, G, N, _ = X_sched.shape
        assert G == self.G
        S, S1, B, D, Dq, dev = self.S, self.S + 1, N // 2, self.D, self.Dq, self.dev
        cand = self.be.candidates(nb * G * N, S, self.item_num, self.seed, epoch).view(nb, G, N, S)
        users = X_sched[:, :, :B, 0]                                                   # rows k and B+k share the user
        items4 = torch.cat([X_sched[:, :, :, 1:2], cand], 3)                           # [nb, G, N, S1] candidate slots (n, s)
        items = items4.reshape(nb, G, N * S1)
        feats = X_sched[:, :, :, 1].contiguous()                                       # true items
        me = self.rank
        re = _Route(torch.cat([users, items], 2), B, self.row_off_v, me, G)            # embedding-side slots: users | items
        rf = _Route(feats, N, 0, me, G)
        T = B + N * S1
        ar = torch.arange(N, device=dev)
        inv_u, inv_i = re.inv[:, :B], re.inv[:, B:].reshape(nb, N, S1)
        Xc = torch.stack([inv_u[:, ar % B], inv_i[:, :, 0]], 2).contiguous()           # compact ids = receive positions
        cand_c = inv_i[:, :, 1:].contiguous()
        # the received feature row j belongs to true-item slot n = perm_f[j]; it must sit at the compact id of item (n, 0)
        feat_dst = torch.gather(inv_i[:, :, 0], 1, rf.perm).to(torch.int32).contiguous()
        f32 = torch.float32
        # [emb | IPS factor | bias | prop] rows — or, with the dense exposure matrix sharded by user rows, [emb | the 2 (S + 1)
        # exposures of the pair this user slot belongs to] (item rows leave that part unused)
        we = D + (2 * S1 if self.dense_expo else Dq + 2)
        e = lambda *shape: torch.empty(shape, dtype=f32, device=dev)
        self.plan = dict(
            nb=nb, N=N, B=B, T=T, re=re, rf=rf, Xc=Xc, cand_c=cand_c, feat_dst=feat_dst,
            send_e=e(re.send_max, we), recv_e=e(T, we), send_f=e(rf.send_max, self.F), recv_f=e(N, self.F),
            E=e(T, D), featc=torch.zeros((T, self.F), dtype=f32, device=dev),
            gc=torch.zeros((T, D), dtype=f32, device=dev), gback=e(re.send_max, D),
            Y=torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)]),
            pred=e(N), loss=e(1))
        p = self.plan
        if self.dense_expo:
            # owner side: the exposures of the pair a sent user row serves — rows k and B + k of rank q's batch, their true item
            # and candidates — gathered from this rank's rows of the matrix, for every step of the epoch in one indexing pass
            na_max = re.a_src.shape[1]
            q_of, k_of = (re.a_slot // T).long(), (re.a_slot % T).long().clamp_(max=B - 1)       # (padding entries: any valid pair)
            stepi = torch.arange(nb, device=dev).view(nb, 1).expand(nb, na_max)
            its = torch.cat([items4[stepi, q_of, k_of], items4[stepi, q_of, k_of + B]], 2)       # [nb, na_max, 2 S1]
            lrow = re.a_src.long().clamp_(max=max(self.nU - 1, 0))
            if self.nU > 0:
                p['ex_all'] = self.expo[lrow.unsqueeze(2), its].reshape(nb * na_max, 2 * S1).contiguous()
            else:
                p['ex_all'] = torch.zeros((max(nb * na_max, 1), 2 * S1), dtype=f32, device=dev)
            p['ex_idx'] = (torch.arange(na_max, device=dev).view(1, na_max) + na_max * torch.arange(nb, device=dev).view(nb, 1)).to(torch.int32).contiguous()
            # receiver side: receive position j carries my slot perm[j]; a user slot k < B delivers [row k's | row B + k's]
            # exposures, an item slot goes to a scratch row
            p['ex_dst'] = torch.where(re.perm < B, re.perm, torch.full_like(re.perm, 2 * B)).to(torch.int32).contiguous()
            p['EG'] = torch.zeros((3 * B + 1, S1), dtype=f32, device=dev)        # rows [0, 2B): Expo[u(n), cand[n][s]] of the batch
            self.tables_u, self.tables_i = [self.U], [self.V]
            p['ipsc'] = None
        else:
            p.update(PQ=e(T, Dq), bb=e(T, 1), prop=e(T, 1))
            self.tables_u = [self.U, self.ips['P'], self.ips['bu'].view(-1, 1), self.user_pad]
            self.tables_i = [self.V, self.ips['Q'], self.ips['bi'].view(-1, 1), self.ips['prop'].view(-1, 1)]
            p['ipsc'] = dict(P=p['PQ'], bu=p['bb'].view(-1), Q=p['PQ'], bi=p['bb'].view(-1), prop=p['prop'].view(-1),
                             b0=self.ips['b0'], M=self.ips['M'])
        # the outgoing payload kinds are ONE launch, the incoming ones (+ zeroing the compact gradient table) another
        z = re.a_src[0] if nb else None
        pack = [(z, z, 0, self.tables_u, p['send_e']), (z, z, 0, self.tables_i, p['send_e']), (z, z, 0, [self.feat], p['send_f'])]
        unpack = [(None, None, T, [p['E']] if self.dense_expo else [p['E'], p['PQ'], p['bb'], p['prop']], p['recv_e']),
                  (None, p['feat_dst'][0] if nb else None, N, [p['featc']], p['recv_f'])]
        if self.dense_expo:
            pack.append((z, z, 0, [p['ex_all']], p['send_e'], D))                 # behind the embedding columns
            unpack.append((None, p['ex_dst'][0] if nb else None, T, [p['EG'], p['EG'][B:]], p['recv_e'], D))
        p['pack'], p['unpack'] = self.be.make_jobs(pack), self.be.make_jobs(unpack)
        if self.lazy_K >= 2 and self.segments and hasattr(self.be, 'lazy_state') and \
                (self.lazy is None or self.lazy.list_cap < re.send_max):
            self.flush()
            self.lazy = self.be.lazy_state(self.flat_p, self.flat_g, self.s1, self.s2, self.lr, self.l2, self.segments, self.lazy_K,
                                           max(G * T, re.send_max), self.opt_name)
            self.lazy.sync_all(self.t)

    # ------------------------------------------------------------------------------------------------ one step
    def _a2a(self, out, inp, out_splits, in_splits, out_np=None, in_np=None):
        if self.comm is not None:
            self.comm.all_to_all_rows(out, inp, in_np, out_np, out.shape[1])
        else:
            dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    def train_step(self, k, marks=None):
        """Step k of the epoch prepared by begin_epoch.  Returns (prediction [2B], loss [1]) of THIS rank's pairs.
        4 collectives: rows (users + candidate items in one payload), feature rows, gradient rows, [dW | db].
        marks: optional callable(name) invoked between the phases (bench: per-phase HIP events)."""
        mark = marks or (lambda name: None)
        p, be = self.plan, self.be
        re, rf = p['re'], p['rf']
        N, T = p['N'], p['T']
        ne, nf = re.send_n[k], rf.send_n[k]
        if self.lazy is not None:        # the rows about to leave (= the rows whose gradients come back) at step t, everything else may lag
            self.lazy.catchup_rows(self.t + 1, re.g_row[k], ne)
            mark('lazy_catchup')
        # rows out: users and items share one payload (both are [embedding | exposure side] rows); one launch
        pk = p['pack']
        be.set_job(pk, 0, re.a_src[k], re.a_dst[k], re.na[k])
        be.set_job(pk, 1, re.b_src[k], re.b_dst[k], re.nb_[k])
        be.set_job(pk, 2, rf.a_src[k], rf.a_dst[k], nf)
        if self.dense_expo:
            be.set_job(pk, 3, p['ex_idx'][k], re.a_dst[k], re.na[k])
        be.pack_multi(pk)
        mark('pack')
        o8 = 8 * self.G * k              # byte offset of step k in the [nb, G] int64 split arrays
        if self.comm is not None:        # both payloads in one RCCL group
            self.comm.all_to_all_rows2(p['recv_e'], p['send_e'], re.send_ad + o8, re.recv_ad + o8,
                                       p['recv_f'], p['send_f'], rf.send_ad + o8, rf.recv_ad + o8)
        else:
            self._a2a(p['recv_e'], p['send_e'][:ne], re.recv_splits[k], re.send_splits[k])
            self._a2a(p['recv_f'], p['send_f'][:nf], rf.recv_splits[k], rf.send_splits[k])
        mark('a2a_rows')
        # compact tables in receive order: ONE table serves users and items (compact id = receive position); the same launch
        # zeroes the step's compact gradient table
        up = p['unpack']
        be.set_job(up, 1, None, p['feat_dst'][k], N)
        if self.dense_expo:
            be.set_job(up, 2, None, p['ex_dst'][k], T)
        be.unpack_multi(up, p['gc'])
        mark('unpack')
        pred, loss = be.local_step(p['E'], p['E'], self.W, self.b, p['featc'], p['ipsc'], p['Xc'][k], p['cand_c'][k], p['Y'],
                                   self.S, self.A, self.std, self.dropout, self.seed, self.word_base + self.t * self.G + self.rank,
                                   p['gc'], p['gc'], self.gW, self.gb, pred=p['pred'], loss=p['loss'],
                                   extra=self.extra, gextra=self.gextra, eg=p['EG'][:N] if self.dense_expo else None)
        # the dense tail [dW | db | extra layers] is complete after the backward: its all-reduce travels while the gradient rows
        # go back to their owners (compact order == receive order: nothing to permute) and are summed there
        mark('fwd_bwd')
        if self.comm is not None:        # (everything on the launch stream, in order: no cross-stream wait to pay for)
            self._a2a(p['gback'][:ne], p['gc'], re.send_splits[k], re.recv_splits[k], re.send_ad + o8, re.recv_ad + o8)
            self.comm.all_reduce_sum(self.g_dense)
            be.scatter_add(re.g_row[k], ne, p['gback'], self.g_rows, self.touched if self.segments else None)
        else:
            work = dist.all_reduce(self.g_dense, group=self.group, async_op=True)
            self._a2a(p['gback'][:ne], p['gc'], re.send_splits[k], re.recv_splits[k])
            be.scatter_add(re.g_row[k], ne, p['gback'], self.g_rows, self.touched if self.segments else None)
            work.wait()
        mark('a2a_grads+all_reduce+scatter')
        if self.lazy is not None:
            self.lazy.opt_step(self.t + 1, ne)
        else:
            be.opt_step(self.flat_p, self.flat_g, self.s1, self.s2, self.lr, self.l2, self.t + 1, self.segments, self.opt_name)
        self.t += 1
        mark('adam')
        return pred, loss

    # ------------------------------------------------------------------------------------------------ first contact with RCCL
    def crosscheck_collectives(self, rows=257, seed=1234):
        """Outside any timed region, before the first step on more than one rank: the same payload through the direct RCCL
        calls (dccf_comm_all_to_all_rows2 / _all_to_all_rows / _all_reduce_sum: code that no one-GPU box can run with a peer) and
        through torch.distributed, compared on the device.  Every rank takes the same decision (the verdict is all-reduced); on
        a mismatch or an error the reason goes to stderr, the direct communicator is closed and the step uses the
        torch.distributed branch.  Returns the name recorded as config.collectives."""
        import sys
        if self.comm is None:
            return self.collectives
        G, dev = self.G, self.dev
        ok, why = 1, ''
        try:
            g = torch.Generator(device='cpu').manual_seed(seed + 17 * self.rank)
            cnt_s = torch.tensor([(rows + 31 * (self.rank + q)) % 97 + 1 for q in range(G)], dtype=torch.int64)
            cnt_r = torch.tensor([(rows + 31 * (q + self.rank)) % 97 + 1 for q in range(G)], dtype=torch.int64)      # what q sends me
            outs = []
            for width in (self.D + 2, self.F):
                send = torch.randn(int(cnt_s.sum()), width, generator=g).to(dev)
                ref = torch.empty(int(cnt_r.sum()), width, device=dev)
                dist.all_to_all_single(ref, send, output_split_sizes=cnt_r.tolist(), input_split_sizes=cnt_s.tolist(), group=self.group)
                outs.append((send, ref, torch.zeros_like(ref)))
            sn, rn = cnt_s.numpy().copy(), cnt_r.numpy().copy()
            (sa, ra, oa), (sb, rb, ob) = outs
            self.comm.all_to_all_rows2(oa, sa, sn.ctypes.data, rn.ctypes.data, ob, sb, sn.ctypes.data, rn.ctypes.data)
            oc = torch.zeros_like(ra)
            self.comm.all_to_all_rows(oc, sa, sn.ctypes.data, rn.ctypes.data, sa.shape[1])
            red = torch.randn(4099, generator=g).to(dev)
            red_ref = red.clone()
            dist.all_reduce(red_ref, group=self.group)
            self.comm.all_reduce_sum(red)
            torch.cuda.synchronize(dev) if dev.type == 'cuda' else None
            if not (torch.equal(oa, ra) and torch.equal(ob, rb) and torch.equal(oc, ra)):
                ok, why = 0, 'all-to-all payloads differ from torch.distributed'
            elif not torch.allclose(red, red_ref, rtol=1e-5, atol=1e-6):
                ok, why = 0, 'all-reduce differs from torch.distributed'
        except Exception as ex:        # noqa: a failing first contact must not take the job down
            ok, why = 0, 'error: %s' % ex
        v = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN, group=self.group)
        if int(v) == 0:
            print('[dccf_amd.sharded] rank %d: direct RCCL collectives failed the cross-check (%s): falling back to torch.distributed'
                  % (self.rank, why or 'a peer reported a mismatch'), file=sys.stderr)
            self.close()
            self.collectives = 'torch.distributed (fallback: direct RCCL cross-check failed%s)' % ((': ' + why) if why else '')
        else:
            self.collectives = 'rccl-direct (cross-checked against torch.distributed)'
        return self.collectives


# ------------------------------------------------------------------------------------------------------ bench entry
def bench_main(args, rank, world, dev):
    """bench.py --gpus N (N > 1): weak scaling — every rank trains `batch_size` pairs per step on its shard."""
    import json
    import time
    from dccf_amd.data_processor import DeviceTrainSet
    import os
    from dccf_amd import utils
    U, I, D, F, B = args.users, args.items, args.dim, args.feat, args.batch_size
    S, A = 10, 2
    # everything up to the end of the warm-up (table setup, the communicator, the first collectives on real RCCL) runs under a
    # host-side deadline: a hang at first contact makes every rank exit non-zero with a message
    deadline = utils.Deadline(float(os.environ.get('DCCF_WARMUP_DEADLINE_S', '420')),
                              'set-up + %d warm-up steps of the row-sharded %d-rank step' % (args.warmup, world))
    be = HipBackend(dev)
    g = torch.Generator(device=dev).manual_seed(args.seed + 1000 * rank)
    nU, nI = (U + world - 1 - rank) // world, (I + world - 1 - rank) // world
    feat = torch.randn(nI, F, generator=g, device=dev) * 0.05
    ips = dict(P=torch.randn(nU, 64, generator=g, device=dev) * 0.1, Q=torch.randn(nI, 64, generator=g, device=dev) * 0.1,
               bu=torch.randn(nU, generator=g, device=dev) * 0.1, bi=torch.randn(nI, generator=g, device=dev) * 0.1,
               prop=torch.rand(nI, generator=g, device=dev), b0=0.1, M=0.1)
    tr = ShardedDCCF(rank, world, U, I, D, S, A, 0.1, 0.2, 1e-3, 1e-4, args.seed, be, dev, feat, ips)
    tr.init_params()
    # first contact with the direct RCCL collectives, outside the timed region: cross-checked against torch.distributed on the
    # device; on a mismatch -> stderr and the torch.distributed branch (config.collectives says which one ran)
    collectives = tr.crosscheck_collectives()
    from bench import synthetic_interactions
    n_steps = max(args.steps, args.warmup) + args.warmup
    n_pairs = (n_steps + 2) * B * world
    uid, iid = synthetic_interactions(int(n_pairs * 1.15) + 1000, U, I, args.seed)      # replicated train set
    ds = DeviceTrainSet(uid[:n_pairs], iid[:n_pairs], U, I, args.seed)

    def schedule(e):
        full, _ = ds.epoch_batches(e, B)                       # same permutation / negatives on every rank
        nb = full.shape[0] // world
        return full[:nb * world].view(nb, world, 2 * B, 2)     # step k: rank r trains full[k*world + r]

    # (the warm-up epoch's plan has the timed epoch's size: the routing tables of the timed one then reuse its allocations)
    tr.begin_epoch(schedule(0)[:max(args.warmup, args.steps)], 0)
    for k in range(args.warmup):
        tr.train_step(k)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    deadline.cancel()
    deadline = utils.Deadline(float(os.environ.get('DCCF_BENCH_DEADLINE_S', '600')), 'the timed %d steps + the per-phase section' % args.steps)
    t0 = time.perf_counter()
    tr.begin_epoch(schedule(1)[:args.steps], 1)                # the epoch's sampling + routing tables are timed
    for k in range(args.steps):
        tr.train_step(k)
    tr.flush()                 # (timed: every row of the shard has received every step when the clock stops)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    # ---- outside the timed region: per-phase GPU time of a step (HIP events between the phases, queue kept full) and its host
    # cost, and the roofline of the layout's HBM-bound kernel — the dense regularised Adam pass over THIS rank's shard
    n_prof = min(args.steps, 50)
    names, evs = [], []
    blocker = torch.zeros(64 << 20, device=dev)
    for _ in range(32):
        blocker.add_(1.0)
    h0 = time.perf_counter()
    for k in range(n_prof):
        row = [torch.cuda.Event(enable_timing=True)]
        row[0].record()

        def mk(name, row=row):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            row.append(e)
            if k == 0:
                names.append(name)
        tr.train_step(k, marks=mk)
        evs.append(row)
    host_us = (time.perf_counter() - h0) / n_prof * 1e6
    torch.cuda.synchronize()
    del blocker
    phase_us = {nm: round(sum(r[i].elapsed_time(r[i + 1]) for r in evs) / n_prof * 1e3, 2) for i, nm in enumerate(names)}
    adam_ms = max(phase_us['adam'] / 1e3, 1e-6)
    n_local = tr.flat_p.numel()
    # the dense regularised Adam pass over THIS rank's shard (what every step runs with lazy_K = 0), timed on its own
    tr.flush()
    evs2 = []
    for k in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        be.opt_step(tr.flat_p, tr.flat_g, tr.s1, tr.s2, tr.lr, tr.l2, tr.t + 1 + k, tr.segments)
        b.record()
        evs2.append((a, b))
    torch.cuda.synchronize()
    dense_ms = sum(a.elapsed_time(b) for a, b in evs2[2:]) / 10
    whole = {'kernel': 'k_dense_opt_rows<Adam> over the local shard (rows mod G)', 'avg_launch_ms': round(dense_ms, 5),
             'algorithmic_GB': round(24.0 * n_local / 1e9, 4), 'achieved': round(24.0 * n_local / 1e9 / (dense_ms / 1e3), 1),
             'frac': round(24.0 * n_local / 1e9 / (dense_ms / 1e3) / 8000.0, 4)}
    if tr.lazy is not None:
        alg = 24.0 * n_local / tr.lazy_K / 1e9
        roofline = {'kernel': 'k_lazy_opt<Adam> over the local shard (windowed lazy regularisation, K = %d)' % tr.lazy_K, 'bound': 'hbm',
                    'achieved': round(alg / (adam_ms / 1e3), 1), 'peak': 8000.0, 'unit': 'GB/s',
                    'frac': round(alg / (adam_ms / 1e3) / 8000.0, 4), 'traffic': None, 'algorithmic_per_launch': round(alg, 4),
                    'avg_launch_ms': round(adam_ms, 5), 'whole_pass': whole,
                    'note': 'the lazy launch is bound by the vector ALU of its replay, not by HBM (DESIGN.md 4b); whole_pass = the '
                            'dense pass over the shard that lazy_K = 0 runs every step; figures include one event boundary (~4 us); '
                            'p + m + v of a shard below 256 MiB sit in the Infinity Cache'}
    else:
        roofline = dict(whole, bound='hbm', peak=8000.0, unit='GB/s', traffic=None, algorithmic_per_launch=whole['algorithmic_GB'],
                        note='includes one event boundary (~4 us); p + m + v of a shard below 256 MiB sit in the Infinity Cache')
    if rank == 0:
        out = {'metric': 'train pairs/sec at rank=64 Electronics', 'value': round(args.steps * B * world / dt, 1),
               'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'DCCF train step (fwd + BPR + bwd + dense l2/clip/Adam), Electronics-shaped synthetic: '
                                      'user_num=%d item_num=%d D=%d F=%d S=%d A=%d, layout=sharded (embedding rows mod %d, all-to-all '
                                      'row exchange, all-reduce of [dW|db]), exposure from IPS factors, fused on-device negatives'
                                      % (U, I, D, F, S, A, world),
                          'layout': 'sharded',
                          'regularisation': ('windowed lazy (K = %d) over the local shard' % tr.lazy_K) if tr.lazy is not None else 'dense pass',
                          'batch_size_per_gpu': B, 'global_batch': B * world, 'optimizer': 'Adam lr=1e-3 l2=1e-4 dropout=0.2',
                          'collectives_per_step': 'all_to_all x3 (rows, feature rows, grad rows) + all_reduce([dW|db])',
                          'collectives': collectives},
               'roofline': roofline, 'phase_us': phase_us, 'host_us_per_step': round(host_us, 1), 'cpu_baseline': None}
        import bench
        bench.emit(out)
    deadline.cancel()
    tr.close()                 # (the directly created communicator goes before the process group does)
    dist.destroy_process_group()
