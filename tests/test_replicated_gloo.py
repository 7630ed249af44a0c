# coding=utf-8
"""CPU, world_size 2, gloo: the replicated data-parallel step (dccf_amd/replicated.py) must equal ONE step on the union of
the ranks' batches, and the replicas must stay bit-identical.  The local compute and the export / import of the touched
gradient rows are played by numpy with the buffer layout of include/dccf_hip.h (dp_export_touched / dp_import_touched),
so this checks the trainer's sequencing, the all-gather and the rank-ordered summation contract."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dccf_oracle as O
from oracle import philox as PH

KEYS = ['uid_embeddings.weight', 'iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias']
CFG = dict(U=37, I=53, D=16, F=32, S=4, A=2, std=0.1, dropout=0.2, lr=0.01, l2=1e-3, seed=77, B=5, steps=3)


class OracleBackend(object):
    """Stands in for replicated.HipBackend: numpy arithmetic, the same Philox streams, the same buffer layout."""

    def local_step(self, tr, X, Y, step, pred):
        P = {KEYS[0]: tr.U.numpy(), KEYS[1]: tr.V.numpy(), KEYS[2]: tr.W.numpy(), KEYS[3]: tr.b.numpy()}
        N = X.shape[0]
        L = N * (tr.S + 1) * tr.A
        cand = PH.candidates(tr.seed, step, N, tr.S, tr.item_num)
        noise = PH.noise(tr.seed, step, L, tr.feat.shape[1], tr.std)
        keep = PH.dropout_keep(tr.seed, step, L, tr.D, float(np.float32(tr.dropout)))
        fw = O.dccf_forward(P, tr.feat.numpy(), tr.expo.numpy(), X.numpy(), cand, noise, keep, tr.dropout, tr.A)
        loss, dpred = O.loss_and_dpred(fw['prediction'], Y.numpy(), 1)
        g = O.dccf_backward(P, fw, dpred, tr.A)
        for dst, k in ((tr.gU, KEYS[0]), (tr.gV, KEYS[1]), (tr.gW, KEYS[2]), (tr.gb, KEYS[3])):
            dst += torch.from_numpy(g[k])
        tr.tU[X[:, 0]] = 1                                        # what the backward kernel flags
        tr.tV[torch.from_numpy(np.concatenate([X[:, 1].numpy(), cand.reshape(-1)]))] = 1
        tr.loss[0] = float(loss)
        return torch.from_numpy(fw['prediction']), tr.loss

    def export(self, tr):
        buf, D, cap = tr.buf, tr.D, tr.cap
        buf.zero_()
        ids, rows = [], []
        for q, (off, n, w, flags) in enumerate(tr.segments):
            for row in torch.nonzero(flags).flatten().tolist():
                ids.append((q << 40) | row)
                rows.append(tr.flat_g[off + row * w:off + (row + 1) * w].clone())
                tr.flat_g[off + row * w:off + (row + 1) * w] = 0
            flags.zero_()
        cnt = len(ids)
        assert cnt <= cap
        buf[:1].view(torch.int32)[0] = cnt
        buf[1] = tr.loss[0]
        buf[4:4 + 2 * cap].view(torch.int64)[:cnt] = torch.tensor(ids, dtype=torch.int64)
        if cnt:
            buf[4 + 2 * cap:4 + 2 * cap + cnt * D] = torch.cat(rows)
        nd = tr.flat_g.numel() - tr.dense_begin
        buf[4 + 2 * cap + cap * D:4 + 2 * cap + cap * D + nd] = tr.flat_g[tr.dense_begin:]
        tr.flat_g[tr.dense_begin:] = 0

    def mark_global(self, tr, X_all, step0):
        """numpy restatement of dp_mark_global: bytes for every row any rank touches."""
        for r in range(X_all.shape[0]):
            X = X_all[r].numpy()
            cand = PH.candidates(tr.seed, step0 + r, X.shape[0], tr.S, tr.item_num)
            tr.gfU[torch.from_numpy(X[:, 0])] = 1
            tr.gfV[torch.from_numpy(np.concatenate([X[:, 1], cand.reshape(-1)]))] = 1

    def opt_untouched(self, tr, t):
        """Phase 1 touches only rows whose global byte is 0; the dense optimizer is element-wise, so the oracle defers the
        arithmetic to opt_touched and checks here what the phase relies on: nothing was (or will be) added to those rows."""
        self.untouched = []
        for off, n, w, flags in tr.gsegments:
            rows = torch.nonzero(flags == 0).flatten()
            assert float(tr.flat_g[off:off + n * w].view(n, w)[rows].abs().max()) == 0.0
            self.untouched.append(rows)

    def opt_touched(self, tr, t):
        for (off, n, w, flags), rows in zip(tr.gsegments, self.untouched):
            assert float(tr.flat_g[off:off + n * w].view(n, w)[rows].abs().max()) == 0.0      # still zero after the import
        assert t == tr.t
        self.opt_step(tr)
        for _, _, _, flags in tr.gsegments:
            flags.zero_()

    # ---- the three calls of a step (interface of replicated.HipBackend), composed from the pieces below
    def local(self, tr, X, Y, step, pred, X_all=None, step0=0):
        pred, _ = self.local_step(tr, X, Y, step, pred)
        self.export(tr)
        if X_all is not None:
            self.mark_global(tr, X_all, step0)
        return pred

    def overlap(self, tr, t):
        self.opt_untouched(tr, t)

    def finish(self, tr, t, ov):
        if ov:
            self.import_apply(tr, t)
        else:
            self.import_(tr)
            self.opt_step(tr)

    def import_apply(self, tr, t):
        self.import_(tr, global_flags=True)
        self.opt_touched(tr, t)

    def import_(self, tr, global_flags=False):
        D, cap, W = tr.D, tr.cap, tr.words
        segments = tr.gsegments if global_flags else tr.segments
        nd = tr.flat_g.numel() - tr.dense_begin
        tr.loss_sum.zero_()
        for r in range(tr.G):                                      # rank order
            b = tr.bufs[r * W:(r + 1) * W]
            cnt = int(b[:1].view(torch.int32)[0])
            ids = b[4:4 + 2 * cap].view(torch.int64)[:cnt].tolist()
            for e, id_ in enumerate(ids):
                q, row = id_ >> 40, id_ & ((1 << 40) - 1)
                off, n, w, flags = segments[q]
                tr.flat_g[off + row * w:off + (row + 1) * w] += b[4 + 2 * cap + e * D:4 + 2 * cap + (e + 1) * D]
                flags[row] = 1
            tr.flat_g[tr.dense_begin:] += b[4 + 2 * cap + cap * D:4 + 2 * cap + cap * D + nd]
            tr.loss_sum += b[1]

    def opt_step(self, tr):
        if not hasattr(self, 'opt'):
            self.opt = O.DenseOptimizer('adam', tr.lr, tr.l2)
        P, _ = O.train_step({'p': tr.flat_p.numpy().copy()}, self.opt, tr.l2, {'p': tr.flat_g.numpy()})
        tr.flat_p.copy_(torch.from_numpy(P['p']))
        tr.flat_g.zero_()
        for _, _, _, flags in tr.segments:
            flags.zero_()


def make_world(c):
    rng = np.random.RandomState(5)
    P = {KEYS[0]: (rng.randn(c['U'], c['D']) * 0.3).astype(np.float32), KEYS[1]: (rng.randn(c['I'], c['D']) * 0.3).astype(np.float32),
         KEYS[2]: (rng.randn(c['D'], c['D'] + c['F']) * 0.1).astype(np.float32), KEYS[3]: (rng.randn(c['D']) * 0.1).astype(np.float32)}
    feat = (rng.randn(c['I'], c['F']) * 0.5).astype(np.float32)
    expo = rng.randn(c['U'], c['I']).astype(np.float32)
    X = []
    for _ in range(c['steps']):
        xs = []
        for _r in range(2):
            u = rng.randint(0, c['U'], c['B'])
            xs.append(np.concatenate([np.stack([u, rng.randint(0, c['I'], c['B'])], 1), np.stack([u, rng.randint(0, c['I'], c['B'])], 1)]))
        X.append(np.stack(xs).astype(np.int64))
    return P, feat, expo, X


def worker(rank, world, port, out, overlap):
    CFG['overlap'] = overlap
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dccf_amd.replicated import ReplicatedDCCF
    c = CFG
    P, feat, expo, X = make_world(c)
    T = torch.from_numpy
    tr = ReplicatedDCCF(rank, world, c['U'], c['I'], c['D'], c['S'], c['A'], c['std'], c['dropout'], c['lr'], c['l2'], c['seed'],
                        OracleBackend(), torch.device('cpu'), T(feat), expo=T(expo), max_rows=2 * c['B'])
    tr.set_params(T(P[KEYS[0]]), T(P[KEYS[1]]), T(P[KEYS[2]]), T(P[KEYS[3]]))
    Y = torch.cat([torch.ones(c['B']), torch.zeros(c['B'])])
    preds, losses = [], []
    for step in range(c['steps']):
        pred, loss = tr.train_step(T(X[step][rank]), Y, X_all=T(X[step]) if c.get('overlap') else None)
        preds.append(pred.numpy().copy())
        losses.append(float(loss))
    np.savez(os.path.join(out, 'rank%d.npz' % rank), p=tr.flat_p.numpy(), U=tr.U.numpy(), V=tr.V.numpy(), W=tr.W.numpy(),
             b=tr.b.numpy(), preds=np.stack(preds), losses=np.array(losses))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('overlap', [False, True])
def test_replicated_step_equals_union_batch(tmp_path, overlap):
    world = 2
    from conftest import free_port
    port = free_port()
    mp.spawn(worker, args=(world, port, str(tmp_path), overlap), nprocs=world, join=True)
    c = CFG
    P, feat, expo, X = make_world(c)
    opt = O.DenseOptimizer('adam', c['lr'], c['l2'])
    N, L = 2 * c['B'], 2 * c['B'] * (c['S'] + 1) * c['A']
    Y = np.concatenate([np.ones(c['B'], np.float32), np.zeros(c['B'], np.float32)])
    res = [dict(np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))) for r in range(world)]
    for step in range(c['steps']):
        total = {k: np.zeros_like(v) for k, v in P.items()}
        loss_sum = 0.0
        for r in range(world):
            st = step * world + r
            cand = PH.candidates(c['seed'], st, N, c['S'], c['I'])
            noise = PH.noise(c['seed'], st, L, c['F'], c['std'])
            keep = PH.dropout_keep(c['seed'], st, L, c['D'], float(np.float32(c['dropout'])))
            fw = O.dccf_forward(P, feat, expo, X[step][r], cand, noise, keep, c['dropout'], c['A'])
            assert np.allclose(res[r]['preds'][step], fw['prediction'], rtol=1e-5, atol=1e-6)
            loss, dpred = O.loss_and_dpred(fw['prediction'], Y, 1)
            loss_sum += float(loss)
            g = O.dccf_backward(P, fw, dpred, c['A'])
            for k in total:
                total[k] += g[k]
        assert np.isclose(res[0]['losses'][step], loss_sum, rtol=1e-5)
        P, _ = O.train_step(P, opt, c['l2'], total)
    for r in range(world):
        for k, name in zip(KEYS, 'UVWb'):
            assert np.allclose(res[r][name], P[k], rtol=1e-5, atol=1e-6), name
    assert np.array_equal(res[0]['p'], res[1]['p'])        # replicas are bit-identical


@pytest.mark.parametrize('n,B,G', [(1000, 64, 2), (1000, 64, 8), (640, 64, 2), (130, 64, 4), (50, 64, 4), (7, 64, 8), (1024, 128, 8)])
def test_epoch_schedule_covers_every_pair_once(n, B, G):
    """replicated.epoch_schedule (what runner._fit_replicated trains under a multi-rank launch): every pair of the epoch is in
    exactly one (step, rank) share, at most G - 1 pairs (the epoch's first) a second time, the shares of a step have equal
    size, and every share keeps the [positives ; negatives] layout with the pair's uid in rows k and b + k
    (src/data_processor/DataProcessor.py:160-207)."""
    from dccf_amd.replicated import epoch_schedule
    rng = np.random.RandomState(n + G)
    uid = rng.randint(0, 500, n)
    pos = np.stack([uid, np.arange(n)], 1)                     # iid = the pair's index: identifies it
    neg = np.stack([uid, 100000 + np.arange(n)], 1)
    nb, r = n // B, n % B
    full = torch.from_numpy(np.stack([np.concatenate([pos[k * B:(k + 1) * B], neg[k * B:(k + 1) * B]]) for k in range(nb)])
                            if nb else np.zeros((0, 2 * B, 2), np.int64))
    tail = torch.from_numpy(np.concatenate([pos[nb * B:], neg[nb * B:]])) if r else None
    sched, last = epoch_schedule(full, tail, G)
    assert sched.shape == (nb // G, G, 2 * B, 2)
    seen = []
    shares = [sched[j, g] for j in range(sched.shape[0]) for g in range(G)]
    if last is not None:
        assert last.shape[0] == G and last.shape[1] % 2 == 0 and last.shape[1] // 2 <= B
        shares += [last[g] for g in range(G)]
    else:
        assert n % (G * B) == 0
    for sh in shares:
        b = sh.shape[0] // 2
        assert torch.equal(sh[:b, 0], sh[b:, 0])                            # the pair shares its uid
        assert torch.equal(sh[b:, 1], sh[:b, 1] + 100000)                   # negative k belongs to positive k
        seen.append(sh[:b, 1].numpy())
    seen = np.concatenate(seen)
    cnt = np.bincount(seen, minlength=n)
    assert cnt.min() == 1 and cnt.sum() - n < G and (cnt[min(B, n):] == 1).all()
    assert len(shares) // G == -(-n // (G * B))


def _crosscheck_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dccf_amd.replicated import ReplicatedDCCF
    c = dict(CFG, overlap=True)
    P, feat, expo, X = make_world(c)
    T = torch.from_numpy
    tr = ReplicatedDCCF(rank, world, c['U'], c['I'], c['D'], c['S'], c['A'], c['std'], c['dropout'], c['lr'], c['l2'], c['seed'],
                        OracleBackend(), torch.device('cpu'), T(feat), expo=T(expo), max_rows=2 * c['B'])
    tr.set_params(T(P[KEYS[0]]), T(P[KEYS[1]]), T(P[KEYS[2]]), T(P[KEYS[3]]))
    Y = torch.cat([torch.ones(c['B']), torch.zeros(c['B'])])
    tr.train_step(T(X[0][rank]), Y, X_all=T(X[0]))
    ok_first = tr.crosscheck_replicas()                 # a healthy warm-up: the replicas agree, nothing changes
    assert ok_first and tr.overlap and 'fallback' not in tr.collectives
    if rank == 1:                                       # what a broken overlapped exchange would leave behind
        tr.flat_p[7] += 1.0
        tr.s1[3] += 0.5
    ok = tr.crosscheck_replicas()
    assert not ok and not tr.overlap and 'fallback' in tr.collectives
    # the job goes on in the synchronous form, from rank 0's state: one more step, then the replicas must still agree
    tr.train_step(T(X[1][rank]), Y, X_all=T(X[1]))
    assert tr.crosscheck_replicas()
    np.savez(os.path.join(out, 'cc%d.npz' % rank), p=tr.flat_p.numpy(), s1=tr.s1.numpy())
    dist.destroy_process_group()


def test_replicas_that_differ_after_warmup_resync_and_fall_back(tmp_path):
    """first contact with real RCCL at G > 1 cannot be rehearsed on one GPU — so the check that guards it is:
    after the warm-up the replicas' checksums are compared; on a mismatch every rank takes rank 0's state, the step falls back
    to its synchronous form and the run says so (config.collectives) instead of timing a drifting job."""
    from conftest import free_port
    mp.spawn(_crosscheck_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (dict(np.load(os.path.join(str(tmp_path), 'cc%d.npz' % r))) for r in range(2))
    assert np.array_equal(a['p'], b['p']) and np.array_equal(a['s1'], b['s1'])
