# coding=utf-8
"""CPU, world_size 2, gloo: the row-sharded training step (dccf_amd/sharded.py) must equal ONE step on the union of
the ranks' batches.  The local compute is played by the oracle (numpy) with the same counter-based draws the HIP
backend uses, so this checks the partitioning, the routing of rows / gradient rows and the dense all-reduce."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dccf_oracle as O
from oracle import philox as PH

KEYS = ['uid_embeddings.weight', 'iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias']
CFG = dict(U=37, I=53, D=16, F=32, Dq=8, S=4, A=2, std=0.1, dropout=0.2, lr=0.01, l2=1e-3, seed=77, B=5, steps=2)


def expo_from_ips(ips):
    e = ips['P'] @ ips['Q'].T + ips['bu'][:, None] + ips['bi'][None, :] + np.float32(ips['b0'])
    return (e / np.maximum(ips['prop'], np.float32(ips['M']))[None, :]).astype(np.float32)


class OracleBackend(object):
    """Stands in for HipBackend: same interface, numpy arithmetic, the same Philox streams."""

    def candidates(self, n_rows, S, item_num, seed, step):
        return torch.from_numpy(PH.candidates(seed, step, n_rows, S, item_num))

    def make_jobs(self, jobs):
        # (idx, dst, n, tables, payload[, col]): a job's rows start at column `col` of the payload rows
        return [dict(idx=j[0], dst=j[1], n=j[2], tables=j[3], payload=j[4][:, (j[5] if len(j) > 5 else 0):]) for j in jobs]

    def set_job(self, jobs, q, idx, dst, n):
        jobs[q].update(idx=idx, dst=dst, n=n)

    def pack_multi(self, jobs):
        for j in jobs:
            n, idx, dst, tables, out = j['n'], j['idx'], j['dst'], j['tables'], j['payload']
            if n:
                rows = torch.cat([t[idx[:n].long()].view(n, -1) for t in tables], 1)
                out[(torch.arange(n) if dst is None else dst[:n].long()), :rows.shape[1]] = rows

    def unpack_multi(self, jobs, zero=None):
        for j in jobs:
            n, dst, payload = j['n'], j['dst'], j['payload']
            rows = torch.arange(n) if dst is None else dst[:n].long()
            o = 0
            for t in j['tables']:
                t[rows] = payload[:n, o:o + t.shape[1]]
                o += t.shape[1]
        if zero is not None:
            zero.zero_()

    def scatter_add(self, idx, n, rows, g, flags=None):
        if n:
            g.index_add_(0, idx[:n].long(), rows[:n])

    def local_step(self, Uc, Vc, W, b, featc, ips, Xc, cand_c, Y, S, A, std, dropout, seed, step, gU, gV, gW, gb, pred=None,
                   loss=None, extra=None, gextra=None, eg=None):
        P = {KEYS[0]: Uc.numpy(), KEYS[1]: Vc.numpy(), KEYS[2]: W.numpy(), KEYS[3]: b.numpy()}
        for k, (wk, bk) in enumerate(extra or []):
            P['mlp.%d.weight' % (k + 1)], P['mlp.%d.bias' % (k + 1)] = wk.numpy(), bk.numpy()
        NL = 1 + len(extra or [])
        N = Xc.shape[0]
        if eg is not None:      # exposures gathered by the owners of the user rows: a matrix over the compact ids that holds them
            T = Uc.shape[0]
            expo = np.zeros((T, T), np.float32)
            items = np.concatenate([Xc.numpy()[:, 1:2], cand_c.numpy()], 1)
            expo[np.repeat(Xc.numpy()[:, 0:1], items.shape[1], 1), items] = eg.numpy()
        else:
            ipn = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in ips.items()}
            expo = expo_from_ips(ipn)
        L = N * (S + 1) * A
        noise = PH.noise(seed, step, L, featc.shape[1], std)
        keep = np.stack([PH.dropout_keep(seed, step, L, Uc.shape[1], float(np.float32(dropout)), layer=k) for k in range(NL)])
        fw = O.dccf_forward(P, featc.numpy(), expo, Xc.numpy(), cand_c.numpy(), noise, keep, dropout, A)
        loss, dpred = O.loss_and_dpred(fw['prediction'], Y.numpy(), 1)
        g = O.dccf_backward(P, fw, dpred, A)
        gU += torch.from_numpy(g[KEYS[0]])
        gV += torch.from_numpy(g[KEYS[1]])
        gW += torch.from_numpy(g[KEYS[2]])
        gb += torch.from_numpy(g[KEYS[3]])
        for k, (gw, gbk) in enumerate(gextra or []):
            gw += torch.from_numpy(g['mlp.%d.weight' % (k + 1)])
            gbk += torch.from_numpy(g['mlp.%d.bias' % (k + 1)])
        return torch.from_numpy(fw['prediction']), torch.tensor([float(loss)])

    def opt_step(self, p, g, s1, s2, lr, l2, t, segments=None, kind='adam'):
        if not hasattr(self, 'opt'):
            self.opt = O.DenseOptimizer(kind, lr, l2)
        P, _ = O.train_step({'p': p.numpy().copy()}, self.opt, l2, {'p': g.numpy()})
        p.copy_(torch.from_numpy(P['p']))
        g.zero_()


def make_world(c):
    rng = np.random.RandomState(5)
    P = {KEYS[0]: (rng.randn(c['U'], c['D']) * 0.3).astype(np.float32), KEYS[1]: (rng.randn(c['I'], c['D']) * 0.3).astype(np.float32),
         KEYS[2]: (rng.randn(c['D'], c['D'] + c['F']) * 0.1).astype(np.float32), KEYS[3]: (rng.randn(c['D']) * 0.1).astype(np.float32)}
    rng2 = np.random.RandomState(11)        # (its own stream: the tables above stay what they were for n_layers = 1)
    for k in range(1, c.get('n_layers', 1)):
        P['mlp.%d.weight' % k] = (rng2.randn(c['D'], c['D']) * 0.3).astype(np.float32)
        P['mlp.%d.bias' % k] = (rng2.randn(c['D']) * 0.1).astype(np.float32)
    feat = (rng.randn(c['I'], c['F']) * 0.5).astype(np.float32)
    ips = dict(P=(rng.randn(c['U'], c['Dq']) * 0.3).astype(np.float32), Q=(rng.randn(c['I'], c['Dq']) * 0.3).astype(np.float32),
               bu=(rng.randn(c['U']) * 0.1).astype(np.float32), bi=(rng.randn(c['I']) * 0.1).astype(np.float32),
               prop=rng.rand(c['I']).astype(np.float32), b0=0.1, M=0.1)
    G, B = 2, c['B']
    X = []
    for _ in range(c['steps']):
        xs = []
        for _r in range(G):
            u = rng.randint(0, c['U'], B)
            xs.append(np.concatenate([np.stack([u, rng.randint(0, c['I'], B)], 1), np.stack([u, rng.randint(0, c['I'], B)], 1)]))
        X.append(np.stack(xs).astype(np.int64))
    return P, feat, ips, X


def build_trainer(c, rank, world, backend, device, P, feat, ips, **kw):
    """The sharded trainer of config c on `device`: exposure as sharded IPS factors (c['expo'] == 'factors') or as this rank's
    rows of the dense matrix those factors give (c['expo'] == 'dense': the same numbers through the other path)."""
    from dccf_amd.sharded import ShardedDCCF
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    ips_loc, expo_loc = None, None
    if c.get('expo', 'factors') == 'dense':
        expo_loc = T(expo_from_ips(ips)[rank::world])
    else:
        ips_loc = dict(P=T(ips['P'][rank::world]), bu=T(ips['bu'][rank::world]), Q=T(ips['Q'][rank::world]),
                       bi=T(ips['bi'][rank::world]), prop=T(ips['prop'][rank::world]), b0=0.1, M=0.1)
    NL = c.get('n_layers', 1)
    tr = ShardedDCCF(rank, world, c['U'], c['I'], c['D'], c['S'], c['A'], c['std'], c['dropout'], c['lr'], c['l2'], c['seed'],
                     backend, device, T(feat[rank::world]), ips_loc, expo_local=expo_loc, opt_name=c.get('opt', 'adam'),
                     n_layers=NL, **kw)
    tr.set_global_params(T(P[KEYS[0]]), T(P[KEYS[1]]), T(P[KEYS[2]]), T(P[KEYS[3]]),
                         extra=[(T(P['mlp.%d.weight' % k]), T(P['mlp.%d.bias' % k])) for k in range(1, NL)])
    return tr


def worker(rank, world, port, out, c):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    P, feat, ips, X = make_world(c)
    tr = build_trainer(c, rank, world, OracleBackend(), torch.device('cpu'), P, feat, ips)
    if c.get('fake_comm'):      # a direct communicator that delivers wrong rows: the cross-check must catch it and fall back
        tr.comm = FakeComm(corrupt=c['fake_comm'] == 'corrupt')
        name = tr.crosscheck_collectives()
        assert (tr.comm is None) == (c['fake_comm'] == 'corrupt') and ('fallback' in name) == (c['fake_comm'] == 'corrupt'), name
    preds = []
    tr.begin_epoch(torch.from_numpy(np.stack(X)), 3)        # [steps, G, 2B, 2], epoch word 3
    for step in range(c['steps']):
        pred, loss = tr.train_step(step)
        preds.append(pred.numpy().copy())
    full_U, full_V = torch.zeros(c['U'], c['D']), torch.zeros(c['I'], c['D'])
    tr.gather_tables(full_U, full_V)
    extra = {'x%d' % k: t.numpy() for k, t in enumerate(x for wb in tr.extra for x in wb)}
    np.savez(os.path.join(out, 'rank%d.npz' % rank), U=tr.U.numpy(), V=tr.V.numpy(), W=tr.W.numpy(), b=tr.b.numpy(),
             preds=np.stack(preds), full_U=full_U.numpy(), full_V=full_V.numpy(), **extra)
    tr.close()
    dist.destroy_process_group()


class FakeComm(object):
    """Stands in for dccf_amd._lib.Comm on the CPU (the real one needs RCCL and a peer GPU): the same three calls over
    torch.distributed, optionally delivering a wrong payload — what a broken first contact with RCCL would look like."""

    def __init__(self, corrupt):
        self.corrupt, self.closed = corrupt, False

    @staticmethod
    def _counts(addr, G):
        import ctypes
        return list((ctypes.c_int64 * G).from_address(addr))

    def all_to_all_rows(self, out, inp, send_rows, recv_rows, width):
        G = dist.get_world_size()
        sr, rr = self._counts(send_rows, G), self._counts(recv_rows, G)
        dist.all_to_all_single(out[:sum(rr)], inp[:sum(sr)].contiguous(), output_split_sizes=rr, input_split_sizes=sr)
        if self.corrupt and dist.get_rank() == 1:
            out[0, 0] += 1.0

    def all_to_all_rows2(self, out_a, inp_a, send_a, recv_a, out_b, inp_b, send_b, recv_b):
        self.all_to_all_rows(out_a, inp_a, send_a, recv_a, out_a.shape[1])
        self.all_to_all_rows(out_b, inp_b, send_b, recv_b, out_b.shape[1])

    def all_reduce_sum(self, buf):
        dist.all_reduce(buf)

    def close(self):
        self.closed = True


VARIANTS = {
    'factors_adam': dict(),
    'dense_adam': dict(expo='dense'),
    'dense_adagrad_l2': dict(expo='dense', opt='adagrad', n_layers=2),
    'factors_gd_l3': dict(opt='gd', n_layers=3, lr=0.05),
    # the directly created communicator (here: a stand-in over gloo) passes / fails its first-contact cross-check
    'comm_ok': dict(fake_comm='ok'),
    'comm_corrupt_falls_back': dict(fake_comm='corrupt'),
}


@pytest.mark.parametrize('variant', sorted(VARIANTS))
def test_sharded_step_equals_union_batch(tmp_path, variant):
    world = 2
    from conftest import free_port
    port = free_port()
    c = dict(CFG, **VARIANTS[variant])
    mp.spawn(worker, args=(world, port, str(tmp_path), c), nprocs=world, join=True)
    P, feat, ips, X = make_world(c)
    expo = expo_from_ips(ips)
    NL = c.get('n_layers', 1)
    opt = O.DenseOptimizer(c.get('opt', 'adam'), c['lr'], c['l2'])
    N, L = 2 * c['B'], 2 * c['B'] * (c['S'] + 1) * c['A']
    Y = np.concatenate([np.ones(c['B'], np.float32), np.zeros(c['B'], np.float32)])
    res = [dict(np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))) for r in range(world)]
    cand_all = PH.candidates(c['seed'], 3, c['steps'] * world * N, c['S'], c['I']).reshape(c['steps'], world, N, c['S'])
    for step in range(c['steps']):
        cand = cand_all[step]
        total = {k: np.zeros_like(v) for k, v in P.items()}
        for r in range(world):
            noise = PH.noise(c['seed'], step * world + r, L, c['F'], c['std'])
            keep = np.stack([PH.dropout_keep(c['seed'], step * world + r, L, c['D'], float(np.float32(c['dropout'])), layer=k)
                             for k in range(NL)])
            fw = O.dccf_forward(P, feat, expo, X[step][r], cand[r], noise, keep, c['dropout'], c['A'])
            assert np.allclose(res[r]['preds'][step], fw['prediction'], rtol=1e-5, atol=1e-6)
            _, dpred = O.loss_and_dpred(fw['prediction'], Y, 1)
            g = O.dccf_backward(P, fw, dpred, c['A'])
            for k in total:
                total[k] += g[k]
        P, _ = O.train_step(P, opt, c['l2'], total)
    for r in range(world):
        assert np.allclose(res[r]['U'], P[KEYS[0]][r::world], rtol=1e-5, atol=1e-6)
        assert np.allclose(res[r]['V'], P[KEYS[1]][r::world], rtol=1e-5, atol=1e-6)
        assert np.allclose(res[r]['W'], P[KEYS[2]], rtol=1e-5, atol=1e-6)
        assert np.allclose(res[r]['b'], P[KEYS[3]], rtol=1e-5, atol=1e-6)
        xs = [res[r]['x%d' % k] for k in range(2 * (NL - 1))]
        for k in range(1, NL):
            assert np.allclose(xs[2 * (k - 1)], P['mlp.%d.weight' % k], rtol=1e-5, atol=1e-6)
            assert np.allclose(xs[2 * (k - 1) + 1], P['mlp.%d.bias' % k], rtol=1e-5, atol=1e-6)
        # gather_tables: every rank ends with the whole tables
        assert np.allclose(res[r]['full_U'], P[KEYS[0]], rtol=1e-5, atol=1e-6) and np.allclose(res[r]['full_V'], P[KEYS[1]], rtol=1e-5, atol=1e-6)
    assert np.allclose(res[0]['W'], res[1]['W'])        # replicas stay identical
