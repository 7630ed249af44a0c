# coding=utf-8
"""GPU: the gradient exchange kernels of the replicated data-parallel path (dp_export_touched / dp_import_touched) and the
trainer at world size 1 over RCCL (the N > 1 logic runs in tests/test_replicated_gloo.py on the CPU)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _process_group_teardown():
    yield
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def make_state(U, I, D, nd_extra, touchedU, touchedV, seed):
    rng = np.random.RandomState(seed)
    sizes = [U * D, I * D, nd_extra]
    pads = [(n + 255) // 256 * 256 for n in sizes]
    g = torch.zeros(sum(pads), device='cuda')
    offs = np.cumsum([0] + pads[:-1])
    tU = torch.zeros((U + 3) // 4 * 4, dtype=torch.uint8, device='cuda')[:U]
    tV = torch.zeros((I + 3) // 4 * 4, dtype=torch.uint8, device='cuda')[:I]
    gU, gV = g[:U * D].view(U, D), g[offs[1]:offs[1] + I * D].view(I, D)
    for t, gv, rows in ((tU, gU, touchedU), (tV, gV, touchedV)):
        rows = torch.as_tensor(rows, device='cuda', dtype=torch.int64)
        t[rows] = 1
        gv[rows] = torch.as_tensor(rng.randn(len(rows), D).astype(np.float32), device='cuda')
    g[offs[2]:] = torch.as_tensor(rng.randn(pads[2]).astype(np.float32), device='cuda')
    segs = [(int(offs[0]), U, D, tU), (int(offs[1]), I, D, tV)]
    return g, segs, int(offs[2]), tU, tV


@pytest.mark.parametrize('D', [64, 16, 128])
def test_export_import_rank_ordered_sum(D):
    """G = 5 simulated ranks with overlapping touched sets: import == sum of the ranks' gradients in rank order, flags
    set exactly on the union, exported gradients zeroed; a second import with the same scratch gives the same result."""
    from dccf_amd import _lib as L
    U, I, G, cap, nd = 1001, 777, 5, 600, 1000
    rng = np.random.RandomState(3)
    words = L.dp_buffer_words(cap, D, 1024)
    bufs = torch.zeros(G * words, device='cuda')
    want = None
    union_u, union_v = set(), set()
    for r in range(G):
        tu = rng.choice(U, 150, replace=False)
        tv = rng.choice(I, 400, replace=False)        # 5 x 400 of 777: heavy overlap between the ranks
        if r == 1:
            tu = np.r_[tu[:-2], [U - 1, U - 2]]       # last rows: the padded flag word
            tu = np.unique(tu)
        union_u |= set(tu.tolist())
        union_v |= set(tv.tolist())
        g, segs, dense_begin, tU, tV = make_state(U, I, D, nd, tu, tv, 10 + r)
        ref = g.clone()
        loss = torch.tensor([1.5 + r], device='cuda')
        L.dp_export_touched(g, segs, dense_begin, loss, bufs[r * words:(r + 1) * words], cap, D)
        torch.cuda.synchronize()
        assert float(g.abs().max()) == 0.0 and int(tU.sum()) == 0 and int(tV.sum()) == 0
        cnt = int(bufs[r * words:r * words + 1].view(torch.int32)[0])
        assert cnt == len(tu) + len(tv)
        want = ref if want is None else want + ref      # rank order: ((r0 + r1) + r2) + ...
    g, segs, dense_begin, tU, tV = make_state(U, I, D, nd, [], [], 99)
    g.zero_()
    loss_sum = torch.zeros(1, device='cuda')
    scratch = L.DpScratch(U + I, G, 'cuda')
    for rep in range(2):                                 # the scratch cleans itself: a second import must work as well
        g.zero_(); tU.zero_(); tV.zero_()
        L.dp_import_touched(bufs, G, g, segs, dense_begin, loss_sum, cap, D, scratch)
        torch.cuda.synchronize()
        assert torch.equal(g, want)                      # bit-exact: same order of additions
        assert int(scratch.mask.abs().sum()) == 0
    assert sorted(torch.nonzero(tU).flatten().tolist()) == sorted(union_u)
    assert sorted(torch.nonzero(tV).flatten().tolist()) == sorted(union_v)
    assert float(loss_sum) == pytest.approx(sum(1.5 + r for r in range(G)))


def test_mark_global_matches_oracle_streams():
    """dp_mark_global: bytes + de-duplicated list == the rows of X_all and of every rank's Philox candidates (oracle)."""
    from dccf_amd import _lib as L
    from oracle import philox as PH
    U, I, G, N, S, seed, step0 = 1003, 517, 3, 40, 10, 77, 12
    rng = np.random.RandomState(0)
    X_all = np.stack([np.stack([rng.randint(0, U, N), rng.randint(0, I, N)], 1) for _ in range(G)]).astype(np.int64)
    fU = torch.zeros((U + 3) // 4 * 4, dtype=torch.uint8, device='cuda')[:U]
    fV = torch.zeros((I + 3) // 4 * 4, dtype=torch.uint8, device='cuda')[:I]
    lst = torch.zeros(G * N * (S + 2) + 8, dtype=torch.int64, device='cuda')
    cnt = torch.tensor([0, 99], dtype=torch.int32, device='cuda')
    L.dp_mark_global(torch.as_tensor(X_all).cuda(), S, I, seed, step0, fU, fV, lst, cnt, 0)
    torch.cuda.synchronize()
    users, items = set(X_all[:, :, 0].reshape(-1).tolist()), set(X_all[:, :, 1].reshape(-1).tolist())
    for r in range(G):
        items |= set(PH.candidates(seed, step0 + r, N, S, I).reshape(-1).tolist())
    assert sorted(torch.nonzero(fU).flatten().tolist()) == sorted(users)
    assert sorted(torch.nonzero(fV).flatten().tolist()) == sorted(items)
    n = int(cnt[0])
    assert n == len(users) + len(items) and int(cnt[1]) == 0          # the other counter is reset for the next step
    got = lst[:n].tolist()
    assert sorted(got) == sorted(list(users) + [(1 << 40) | i for i in items])


def test_mark_global_without_list_and_counters():
    """The public C entry with NULL list / counters (bytes only — how the replicated step marks an unannounced step's rows): round 2
    found k_dp_mark writing `*cnt_next = 0` through the NULL counter there (a GPU abort, fixed in 9fc7d5d); this calls the entry
    that way directly.  The bytes must be exactly the rows of X_all and of every rank's Philox candidates."""
    import ctypes as C
    from dccf_amd import _lib as L
    from oracle import philox as PH
    U, I, G, N, S, seed, step0 = 1003, 517, 3, 40, 10, 77, 12
    rng = np.random.RandomState(1)
    X_all = np.stack([np.stack([rng.randint(0, U, N), rng.randint(0, I, N)], 1) for _ in range(G)]).astype(np.int64)
    Xd = torch.as_tensor(X_all).cuda()
    fU = torch.zeros((U + 3) // 4 * 4, dtype=torch.uint8, device='cuda')[:U]
    fV = torch.zeros((I + 3) // 4 * 4, dtype=torch.uint8, device='cuda')[:I]
    L.check(L.load().dp_mark_global(L.ptr(Xd, torch.int64), G, N, S, I, seed, step0, L.ptr(fU, torch.uint8), L.ptr(fV, torch.uint8),
                                    0, 1, None, None, None, L.stream()))
    torch.cuda.synchronize()
    users, items = set(X_all[:, :, 0].reshape(-1).tolist()), set(X_all[:, :, 1].reshape(-1).tolist())
    for r in range(G):
        items |= set(PH.candidates(seed, step0 + r, N, S, I).reshape(-1).tolist())
    assert sorted(torch.nonzero(fU).flatten().tolist()) == sorted(users)
    assert sorted(torch.nonzero(fV).flatten().tolist()) == sorted(items)


@pytest.mark.parametrize('overlap', [False, True, 'prep'])
def test_replicated_trainer_world1_equals_single_gpu_step(overlap):
    """At G = 1 the replicated step (fwd/bwd -> export -> [all_gather] -> import -> Adam), with and without the optimizer
    pass over the untouched rows beside the collective, and with every step prepared by the one before ('prep': no k_prep,
    list export, marks one step ahead), must equal the plain step."""
    prep, overlap = overlap == 'prep', bool(overlap)
    import torch.distributed as dist
    from dccf_amd import replicated, _lib as L
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    from conftest import free_port
    os.environ['MASTER_PORT'] = str(free_port())
    dev = torch.device('cuda', 0)
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    # (the process group is left alive for the second parametrisation; pytest tears the process down)
    U, I, D, F, S, A, B = 500, 300, 64, 160, 10, 2, 48
    g = torch.Generator(device='cuda').manual_seed(1)
    feat = torch.randn(I, F, generator=g, device='cuda') * 0.3
    expo = torch.randn(U, I, generator=g, device='cuda')
    tr = replicated.ReplicatedDCCF(0, 1, U, I, D, S, A, 0.1, 0.2, 1e-3, 1e-4, 5, replicated.HipBackend(dev), dev, feat,
                                   expo=expo, max_rows=2 * B, overlap=overlap)
    tr.init_params(0.1)
    # the plain path on a copy
    p0 = tr.flat_p.clone()
    gg, s1, s2 = torch.zeros_like(p0), torch.zeros_like(p0), torch.zeros_like(p0)
    views = [p0[o:o + n * w].view(n, w) for (o, n, w, _) in tr.segments]
    oW = tr.dense_begin
    W, b = p0[oW:oW + D * (D + F)].view(D, D + F), p0[oW + D * (D + F):oW + D * (D + F) + D]
    gviews = [gg[o:o + n * w].view(n, w) for (o, n, w, _) in tr.segments]
    gW, gb = gg[oW:oW + D * (D + F)].view(D, D + F), gg[oW + D * (D + F):oW + D * (D + F) + D]
    ctx = L.Context(0)
    y = torch.cat([torch.ones(B, device='cuda'), torch.zeros(B, device='cuda')])
    gen = torch.Generator(device='cuda').manual_seed(2)
    seen = []
    T = 5
    sched = torch.stack([torch.stack([torch.randint(0, U // 2, (2 * B,), generator=gen, device='cuda'),
                                      torch.randint(0, I, (2 * B,), generator=gen, device='cuda')], 1) for _ in range(T)])[:, None]
    moved = sched[T - 1].clone()      # the last step's batch at ANOTHER address: what step T-2 prepared must be discarded
    for t in range(T):
        cur = moved if (prep and t == T - 1) else sched[t]
        X = cur[0]
        seen.append(X[:, 0])
        pred, loss = tr.train_step(X, y, X_all=cur if overlap else None,
                                   X_all_next=sched[t + 1] if prep and t + 1 < T and t != 2 else None)   # (one gap)
        m = L.model_struct(views[0], views[1], W, b, feat, expo, S, A, 0.1)
        pred2, loss2 = L.dccf_train_fwdbwd(ctx, m, L.rand_struct(seed=5, step=t), X, y, 1, 0.2, gviews[0], gviews[1], gW, gb)
        L.dense_opt_step('adam', p0, gg, s1, s2, 1e-3, 1e-4, 1e-4, 50.0, t + 1)
        torch.cuda.synchronize()
        np.testing.assert_allclose(pred.cpu().numpy(), pred2.cpu().numpy(), rtol=1e-5, atol=1e-6)
        assert float(loss) == pytest.approx(float(loss2), rel=1e-5)
    assert (tr.be.lazy is not None) == overlap           # the overlapped step runs the windowed lazy regularisation
    tr.flush()
    # float atomics inside the backward reorder sums between the two runs; Adam turns that into a fraction of lr
    d = (tr.flat_p - p0).abs()
    assert float(d.max()) <= 4 * 1e-3 and int((d > 4 * 5e-3 * 1e-3).sum()) <= 4 * D + 8
    assert int(tr.tU.sum()) == 0 and int(tr.tV.sum()) == 0 and float(tr.flat_g.abs().max()) == 0.0
    assert int(tr.gfU.sum()) == 0 and int(tr.gfV.sum()) == 0
    assert sum(int(a.sum()) for a in (tr.gfU2, tr.gfV2, tr.lfU, tr.lfV)) == 0
    assert int(tr.pmask.abs().sum()) == 0 and int((tr.pwhere != 0x7fffffff).sum()) == 0      # import tables consumed
    assert tr.be.ctx.prepared_steps() == (2 if prep else 0)      # steps 1 and 2; step 3 follows the gap, step 4 the mismatch
    # rows no batch touched are bit-identical (their update never sees a float atomic)
    never = torch.ones(U, dtype=torch.bool, device='cuda')
    never[torch.cat(seen)] = False
    assert torch.equal(tr.U[never], views[0][never])
    torch.cuda.synchronize()


def _rank_main(rank, world, port, out, overlap, opt_name='adam', n_layers=1, D=64):
    import torch.distributed as dist
    tag = str(overlap)
    prep, overlap = overlap == 'prep', bool(overlap)
    from dccf_amd import replicated
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)       # two ranks share the one GPU of the test box
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    c = dict(W2, D=D)
    g = torch.Generator(device='cuda').manual_seed(1)
    feat = torch.randn(c['I'], c['F'], generator=g, device='cuda') * 0.3
    expo = torch.randn(c['U'], c['I'], generator=g, device='cuda')
    tr = replicated.ReplicatedDCCF(rank, world, c['U'], c['I'], c['D'], c['S'], c['A'], 0.1, 0.2, 1e-2, 1e-4, 5,
                                   replicated.HipBackend(dev), dev, feat, expo=expo, max_rows=2 * c['B'], overlap=overlap,
                                   opt_name=opt_name, n_layers=n_layers)
    tr.init_params(0.1)
    y = torch.cat([torch.ones(c['B'], device='cuda'), torch.zeros(c['B'], device='cuda')])
    gen = torch.Generator(device='cuda').manual_seed(2)
    losses = []
    sched = torch.stack([torch.stack([torch.stack([torch.randint(0, c['U'], (2 * c['B'],), generator=gen, device='cuda'),
                                                   torch.randint(0, c['I'], (2 * c['B'],), generator=gen, device='cuda')], 1)
                                      for _ in range(world)]) for _ in range(c['steps'])])
    for t in range(c['steps']):
        _, loss = tr.train_step(sched[t, rank], y, X_all=sched[t] if overlap else None,
                                X_all_next=sched[t + 1] if prep and t + 1 < c['steps'] else None)
        losses.append(float(loss))
    tr.flush()
    torch.cuda.synchronize()
    assert tr.be.ctx.prepared_steps() == (c['steps'] - 1 if prep else 0)
    np.savez(os.path.join(out, 'r%d_%s.npz' % (rank, tag)), p=tr.flat_p.cpu().numpy(), losses=np.array(losses),
             flags=np.array([int(a.sum()) for a in (tr.tU, tr.tV, tr.gfU, tr.gfV, tr.gfU2, tr.gfV2, tr.lfU, tr.lfV)]
                            + [int(tr.pmask.abs().sum()), int((tr.pwhere != 0x7fffffff).sum())]),
             gmax=float(tr.flat_g.abs().max()))
    dist.destroy_process_group()


W2 = dict(U=700, I=450, D=64, F=96, S=10, A=2, B=40, steps=5)


@pytest.mark.parametrize('overlap,world,opt_name,n_layers,D', [
    (False, 2, 'adam', 1, 64), (True, 2, 'adam', 1, 64), ('prep', 2, 'adam', 1, 64), ('prep', 3, 'adam', 1, 64),
    ('prep', 2, 'adagrad', 1, 64), ('prep', 2, 'gd', 1, 64), (False, 2, 'gd', 1, 64), ('prep', 2, 'adam', 2, 64),
    (True, 2, 'adam', 3, 64), (False, 2, 'adagrad', 2, 64), ('prep', 2, 'adam', 1, 16), ('prep', 2, 'adam', 1, 128),
    ('prep', 2, 'adam', 1, 48), (True, 2, 'adam', 1, 24), (False, 2, 'gd', 1, 100), ('prep', 2, 'adagrad', 2, 100)])
def test_two_ranks_hip_backend_replicas_bit_identical(tmp_path, overlap, world, opt_name, n_layers, D):
    """World size 2 (and 3) with the HIP backend (the ranks share this box's one GPU, gloo as the transport): the replicas
    end bit-identical, nothing is left in the gradient buffer, the flags or the tables, and the result equals the same
    batches accumulated into one gradient on one GPU, to the float-atomic tolerance."""
    import torch.multiprocessing as mp
    from dccf_amd import _lib as L
    from conftest import free_port
    port = free_port()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path), overlap, opt_name, n_layers, D), nprocs=world, join=True)
    rs = [dict(np.load(os.path.join(str(tmp_path), 'r%d_%s.npz' % (r, str(overlap))))) for r in range(world)]
    r0 = rs[0]
    for r1 in rs[1:]:
        assert np.array_equal(r0['p'], r1['p'])
        assert np.array_equal(r0['losses'], r1['losses'])
    assert all(r['flags'].sum() == 0 and r['gmax'] == 0 for r in rs)
    # one GPU, both batches into one gradient, one optimizer step
    c = dict(W2, D=D)
    from dccf_amd import replicated
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cuda').manual_seed(1)
    feat = torch.randn(c['I'], c['F'], generator=g, device='cuda') * 0.3
    expo = torch.randn(c['U'], c['I'], generator=g, device='cuda')
    tr = replicated.ReplicatedDCCF(0, 1, c['U'], c['I'], c['D'], c['S'], c['A'], 0.1, 0.2, 1e-2, 1e-4, 5,
                                   replicated.HipBackend(dev), dev, feat, expo=expo, max_rows=2 * c['B'], n_layers=n_layers)
    tr.init_params(0.1)
    ctx = L.Context(0)
    y = torch.cat([torch.ones(c['B'], device='cuda'), torch.zeros(c['B'], device='cuda')])
    gen = torch.Generator(device='cuda').manual_seed(2)
    s1 = torch.zeros_like(tr.flat_p) if opt_name != 'gd' else None
    s2 = torch.zeros_like(tr.flat_p) if opt_name == 'adam' else None
    for t in range(c['steps']):
        X_all = torch.stack([torch.stack([torch.randint(0, c['U'], (2 * c['B'],), generator=gen, device='cuda'),
                                          torch.randint(0, c['I'], (2 * c['B'],), generator=gen, device='cuda')], 1)
                             for _ in range(world)])
        total = 0.0
        for r in range(world):
            m = L.model_struct(tr.U, tr.V, tr.W, tr.b, feat, expo, c['S'], c['A'], 0.1, extra=tr.extra)
            _, loss = L.dccf_train_fwdbwd(ctx, m, L.rand_struct(seed=5, step=t * world + r), X_all[r].contiguous(), y, 1, 0.2,
                                          tr.gU, tr.gV, tr.gW, tr.gb, gextra=tr.gextra)
            total += float(loss)
        assert total == pytest.approx(float(r0['losses'][t]), rel=1e-4)
        L.dense_opt_step(opt_name, tr.flat_p, tr.flat_g, s1, s2, 1e-2, 1e-4, 1e-4, 50.0, t + 1)
    d = np.abs(tr.flat_p.cpu().numpy() - r0['p'])
    assert d.max() <= c['steps'] * 1e-2 and (d > c['steps'] * 5e-3 * 1e-2).sum() <= 4 * c['D'] + 8


# ------------------------------------------------------------------------------------------------ row-sharded layout, 2 ranks
SHARD_CASES = {
    # the round-2 case: D = 16, F = 32, 8-d IPS factors
    'small': dict(steps=5),
    # BASELINE.json config 5's model shape: rank-128 embeddings, the 768-d feature projection, 64-d IPSBiasedMF factors for the
    # exposure (a dense U x I matrix is impossible at 10M x 1M), S = 10 candidates, A = 2 noise draws
    'config5': dict(U=301, I=257, D=128, F=768, Dq=64, S=10, A=2, B=16, steps=5),
    # the dense exposure file sharded by user rows (what the CLI has), Adagrad, --n_layers 2
    'dense_adagrad_l2': dict(U=211, I=157, D=64, F=96, S=10, A=2, B=12, steps=5, expo='dense', opt='adagrad', n_layers=2),
    # an embedding size that is not a kernel tile (no row segments: the dense optimizer pass over the shard), GD, 3 layers
    'gd_d48_l3': dict(U=97, I=83, D=48, F=100, S=4, A=2, B=9, steps=4, expo='dense', opt='gd', n_layers=3, lr=0.05),
}


def _sharded_rank_main(rank, world, port, out, lazy_K, case):
    """One rank of the row-sharded trainer with the HIP backend; both ranks share this box's one GPU, gloo moves the bytes."""
    import torch.distributed as dist
    import test_sharded_gloo as TS
    from dccf_amd import sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    c = dict(TS.CFG, **SHARD_CASES[case])      # (five steps: the lazy window cycles at K = 2)
    P, feat, ips, X = TS.make_world(c)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tr = TS.build_trainer(c, rank, world, sharded.HipBackend(dev), dev, P, feat, ips, lazy_K=lazy_K)
    assert tr.crosscheck_collectives() == 'torch.distributed'          # (gloo: there is no direct communicator to check)
    preds, losses = [], []
    tr.begin_epoch(T(np.stack(X)), 3)
    for step in range(c['steps']):
        pred, loss = tr.train_step(step)
        preds.append(pred.cpu().numpy().copy())
        losses.append(float(loss))
    assert (tr.lazy is not None) == (lazy_K >= 2 and c['D'] % 4 == 0)
    tr.flush()                 # rows the lazy regularisation left behind are brought up to date before anybody looks
    torch.cuda.synchronize()
    extra = {'x%d' % k: t.cpu().numpy() for k, t in enumerate(x for wb in tr.extra for x in wb)}
    np.savez(os.path.join(out, 'sh%d.npz' % rank), U=tr.U.cpu().numpy(), V=tr.V.cpu().numpy(), W=tr.W.cpu().numpy(),
             b=tr.b.cpu().numpy(), preds=np.stack(preds), losses=np.array(losses), gmax=float(tr.flat_g.abs().max()),
             touched=int(tr.touched.sum()), **extra)
    tr.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('case,lazy_K', [('small', 0), ('small', 2), ('small', 8), ('config5', 0), ('config5', 8), ('config5', 2),
                                         ('dense_adagrad_l2', 2), ('dense_adagrad_l2', 0), ('gd_d48_l3', 8)])
def test_two_ranks_hip_backend_row_sharded_equals_union_batch(tmp_path, lazy_K, case):
    """World size 2 of the ROW-SHARDED layout with the HIP backend (pack / all-to-all / unpack / the single-GPU kernels on
    compact tables / all-to-all / scatter-add / all-reduce / row-aware optimizer — dense, or the windowed lazy regularisation of
    the shard: the rows about to be sent are caught up first): every rank's predictions equal the oracle's on
    the same counter-based draws, the shards partition the tables (rank r holds rows r, r + G, ...), W, b and the extra layers
    stay replicated, and the parameters equal ONE oracle step on the union of the ranks' batches to the float-atomic tolerance.
    Cases: BASELINE config 5's model shape (D = 128, F = 768, 64-d IPS factors, S = 10, A = 2); the dense exposure matrix
    sharded by user rows with Adagrad and --n_layers 2; GD with an embedding size that is not a kernel tile and 3 layers."""
    import torch.multiprocessing as mp
    import test_sharded_gloo as TS
    from oracle import dccf_oracle as O
    from oracle import philox as PH
    world = 2
    from conftest import free_port
    port = free_port()
    mp.spawn(_sharded_rank_main, args=(world, port, str(tmp_path), lazy_K, case), nprocs=world, join=True)
    c = dict(TS.CFG, **SHARD_CASES[case])
    K = TS.KEYS
    P, feat, ips, X = TS.make_world(c)
    expo = TS.expo_from_ips(ips)
    NL, opt_name = c.get('n_layers', 1), c.get('opt', 'adam')
    opt = O.DenseOptimizer(opt_name, c['lr'], c['l2'])
    N, Ld = 2 * c['B'], 2 * c['B'] * (c['S'] + 1) * c['A']
    Y = np.concatenate([np.ones(c['B'], np.float32), np.zeros(c['B'], np.float32)])
    res = [dict(np.load(os.path.join(str(tmp_path), 'sh%d.npz' % r))) for r in range(world)]
    cand_all = PH.candidates(c['seed'], 3, c['steps'] * world * N, c['S'], c['I']).reshape(c['steps'], world, N, c['S'])
    for step in range(c['steps']):
        total = {k: np.zeros_like(v) for k, v in P.items()}
        for r in range(world):
            noise = PH.noise(c['seed'], step * world + r, Ld, c['F'], c['std'])
            keep = np.stack([PH.dropout_keep(c['seed'], step * world + r, Ld, c['D'], float(np.float32(c['dropout'])), layer=k)
                             for k in range(NL)])
            fw = O.dccf_forward(P, feat, expo, X[step][r], cand_all[step][r], noise, keep, c['dropout'], c['A'])
            if step == 0:      # later steps start from parameters that differ by the float-atomic noise of the step before
                np.testing.assert_allclose(res[r]['preds'][step], fw['prediction'], rtol=1e-4, atol=2e-6 * max(1.0, np.abs(fw['prediction']).max()))
            loss, dpred = O.loss_and_dpred(fw['prediction'], Y, 1)
            assert float(loss) == pytest.approx(float(res[r]['losses'][step]), rel=2e-3)
            g = O.dccf_backward(P, fw, dpred, c['A'])
            for k in total:
                total[k] += g[k]
        P, _ = O.train_step(P, opt, c['l2'], total)
    lr, steps = c['lr'], c['steps']
    for r in range(world):
        assert res[r]['gmax'] == 0.0 and res[r]['touched'] == 0            # nothing left behind
        assert res[r]['U'].shape[0] == len(range(r, c['U'], world)) and res[r]['V'].shape[0] == len(range(r, c['I'], world))
        pairs = [(res[r]['U'], P[K[0]][r::world]), (res[r]['V'], P[K[1]][r::world]), (res[r]['W'], P[K[2]]), (res[r]['b'], P[K[3]])]
        for k in range(1, NL):
            pairs += [(res[r]['x%d' % (2 * (k - 1))], P['mlp.%d.weight' % k]), (res[r]['x%d' % (2 * (k - 1) + 1)], P['mlp.%d.bias' % k])]
        for mine, ref in pairs:
            d = np.abs(mine - ref)
            if opt_name == 'gd':       # no amplification: plain fp32 agreement
                assert d.max() <= 3e-5 * max(1.0, np.abs(ref).max()), d.max()
            else:      # Adam / Adagrad turn the last-bit differences of float-atomic sums into fractions of lr on a few elements
                assert d.max() <= steps * lr and (d > steps * 5e-3 * lr).mean() <= 0.02, (d.max(), (d > steps * 5e-3 * lr).mean())
    assert np.array_equal(res[0]['W'], res[1]['W']) and np.array_equal(res[0]['b'], res[1]['b'])     # replicas stay identical
    for k in range(2 * (NL - 1)):
        assert np.array_equal(res[0]['x%d' % k], res[1]['x%d' % k])


# ------------------------------------------------------------------------------------------------ the CLI on two ranks
CLI_SEEDS = [2019 + i for i in range(int(os.environ.get('DCCF_TEST_CLI_SEEDS', '6')))]      # (41 = every seed the reference was run with)


def _cli_rank_main(rank, world, port, tmp, cfg):
    """One rank of `python -m torch.distributed.run --nproc-per-node 2 -m dccf_amd.main ...` (the environment the launcher
    sets, gloo as the transport: both ranks share this box's one GPU), once per seed."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), DCCF_DIST_BACKEND='gloo')
    os.environ.update(cfg.get('env', {}))
    from dccf_amd import main as M
    os.makedirs(os.path.join(tmp, 'src'), exist_ok=True)
    os.chdir(os.path.join(tmp, 'src'))
    valid, init, out = [], [], {}
    for seed in cfg.get('seeds', CLI_SEEDS):
        runner = M.main(['--rank', '1', '--model_name', 'DCCF', '--optimizer', cfg.get('optimizer', 'Adam'), '--lr', str(cfg['lr']), '--dataset', 'toy',
                         '--path', '../dataset/', '--metric', 'ndcg@5,recall@5,auc', '--epoch', str(cfg['epochs']),
                         '--test_neg_n', str(cfg['test_neg_n']), '--u_vector_size', str(cfg['D']), '--i_vector_size', str(cfg['D']),
                         '--batch_size', str(cfg['batch_size'] // world), '--check_epoch', '0', '--random_seed', str(seed),
                         '--verbose', '30', '--model_path', '../model/DCCF/two.pt'] + cfg.get('more', []))
        valid.append(np.array(runner.valid_results))
        init.append(np.array(runner.init_results[1]))
        m = runner.model
        out = dict(t=m.optimizer.t, call=m._call, **{k: v.cpu().numpy() for k, v in m.state_dict().items()})
    torch.cuda.synchronize()
    np.savez(os.path.join(tmp, 'cli%d.npz' % rank), valid=np.stack(valid), init=np.stack(init), **out)
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize('layout', ['replicated', 'sharded'])
def test_cli_on_two_ranks(tmp_path, layout):
    """(layout: --mp replicated, the default, or --mp sharded — embedding rows, optimizer state, feature rows and the rows of the
    dense exposure file sharded by row mod 2, all-to-all row exchange, tables gathered back for the evaluation.)
    dccf_amd.main under a two-rank launch (src/main.py:24-195 is single-GPU: new capability behind the same flags), on the
    dataset and hyper-parameters of the reference's own main.py runs (tests/golden/e2e.npz) with --batch_size halved: every
    optimizer step trains the two ranks' batches as ONE step, i.e. the reference's step at its batch size — the seed-averaged
    validation NDCG@5 of every epoch must agree with the reference's runs like the single-GPU CLI does.  The replicas —
    parameters, metrics of every epoch, step and Philox counters — are identical, rank 0 alone wrote the checkpoint / rank.csv /
    result files, and an epoch whose batches do not divide by the world size ends in one step of equal shares."""
    import torch.multiprocessing as mp
    import pandas as pd
    from conftest import load_golden
    from dccf_amd import synth
    g = load_golden('e2e')
    tmp = str(tmp_path)
    synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', int(g['user_num']), int(g['item_num']), int(g['n_draws']),
                        feat_dim=int(g['feat_dim']), seed=int(g['data_seed']))
    cfg = dict(lr=float(g['lr']), epochs=int(g['epochs']), test_neg_n=int(g['test_neg_n']), D=int(g['D']),
               batch_size=int(g['batch_size']), more=['--mp', layout])
    from conftest import free_port
    port = free_port()
    mp.spawn(_cli_rank_main, args=(2, port, tmp, cfg), nprocs=2, join=True)
    r0, r1 = (dict(np.load(os.path.join(tmp, 'cli%d.npz' % r))) for r in range(2))
    for k in r0:
        assert np.array_equal(r0[k], r1[k], equal_nan=True), k
    n_pairs = len(pd.read_csv(os.path.join(tmp, 'dataset', 'toy', 'toy.train.csv'), header=None))
    B = cfg['batch_size']
    assert n_pairs % B != 0 and int(r0['t']) == cfg['epochs'] * ((n_pairs + B - 1) // B)
    files = [os.path.join(r, f) for r, _, fs in os.walk(tmp) for f in fs]
    assert sum(f.endswith('two.pt') for f in files) == 1 and sum(f.endswith('rank.csv') for f in files) == 1
    assert sum(f.endswith('.npy') and os.sep + 'result' + os.sep in f for f in files) == len(CLI_SEEDS)
    sd = torch.load(os.path.join(tmp, 'model', 'DCCF', 'two.pt'), map_location='cpu')
    assert set(sd) == {'uid_embeddings.weight', 'iid_embeddings.weight', 'mlp.0.weight', 'mlp.0.bias'}
    # statistical parity with the reference's runs (same test as tests/test_e2e_gpu.py, fewer seeds on this side)
    seeds = [int(s) for s in g['seeds']]
    ref = np.stack([g['seed%d/valid' % s][:, 0] for s in seeds])
    mine = r0['valid'][:, :, 0]
    assert mine.shape == (len(CLI_SEEDS), cfg['epochs']) and np.isfinite(r0['valid']).all()
    if os.environ.get('DCCF_TEST_DUMP'):
        import json
        with open(os.environ['DCCF_TEST_DUMP'], 'w') as f:
            json.dump({'two_ranks_seeds': len(CLI_SEEDS), 'reference_seeds': len(seeds), 'batch_size_per_rank': B // 2,
                       'two_ranks_mean': mine.mean(0).tolist(), 'two_ranks_se': (mine.std(0, ddof=1) / len(CLI_SEEDS) ** 0.5).tolist(),
                       'reference_mean': ref.mean(0).tolist(), 'reference_se': (ref.std(0, ddof=1) / len(seeds) ** 0.5).tolist()}, f)
    for e in range(cfg['epochs']):
        se = np.sqrt(ref[:, e].var(ddof=1) / len(seeds) + mine[:, e].var(ddof=1) / len(CLI_SEEDS))
        assert abs(mine[:, e].mean() - ref[:, e].mean()) <= 3 * se + 2e-3, \
            'epoch %d: two ranks %.4f vs reference %.4f (se %.4f)' % (e + 1, mine[:, e].mean(), ref[:, e].mean(), se)


@pytest.mark.parametrize('D', [16, 24])
@pytest.mark.parametrize('layout', ['replicated', 'sharded'])
def test_cli_on_two_ranks_extra_layers_adagrad(tmp_path, layout, D):
    """The same launch (both layouts) with --n_layers 2 --optimizer Adagrad and a batch size that leaves the epoch's schedule uneven: the extra
    layers travel in the dense tail of the exchange, the replicas stay identical and the checkpoint holds all six tensors.  D = 24: an
    embedding size that is no column tile (src/models/RecModel.py:17-27 accepts any)."""
    import torch.multiprocessing as mp
    from dccf_amd import synth
    tmp = str(tmp_path)
    synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', 300, 200, 5000, feat_dim=32, seed=3)
    cfg = dict(lr=0.01, epochs=2, test_neg_n=50, D=D, batch_size=96, seeds=[7], optimizer='Adagrad', more=['--n_layers', '2', '--mp', layout])
    from conftest import free_port
    mp.spawn(_cli_rank_main, args=(2, free_port(), tmp, cfg), nprocs=2, join=True)
    r0, r1 = (dict(np.load(os.path.join(tmp, 'cli%d.npz' % r))) for r in range(2))
    for k in r0:
        assert np.array_equal(r0[k], r1[k], equal_nan=True), k
    assert {'mlp.1.weight', 'mlp.1.bias'} <= set(r0) and r0['mlp.1.weight'].shape == (D, D)
    sd = torch.load(os.path.join(tmp, 'model', 'DCCF', 'two.pt'), map_location='cpu')
    assert len(sd) == 6
    # the extra layer was trained: N(0, 0.01) at the start, Adagrad's first steps move every touched element by ~lr
    assert np.abs(r0['mlp.1.weight']).max() > 0.02 and np.isfinite(r0['valid']).all()


def test_cli_two_ranks_sharded_evaluation_equals_unsharded(tmp_path):
    """Under a multi-rank launch the evaluation batches are dealt to the ranks and the predictions meet in one sum-all-reduce
    (runner.predict_device); every batch keeps the Philox step the single-process loop gives it, so the metrics of every epoch, the
    untrained model's and the call counter equal those of the run in which every rank evaluates everything (DCCF_SHARD_EVAL=0).  (src/runners/BaseRunner.py:134-157, 305-332 evaluate on one device.)"""
    import torch.multiprocessing as mp
    from dccf_amd import synth
    from conftest import free_port
    res = {}
    for mode in ('1', '0'):
        tmp = os.path.join(str(tmp_path), 'm' + mode)
        synth.write_dataset(os.path.join(tmp, 'dataset'), 'toy', 300, 200, 5000, feat_dim=32, seed=3)
        cfg = dict(lr=0.01, epochs=2, test_neg_n=50, D=16, batch_size=96, seeds=[7], optimizer='Adam', more=[], env={'DCCF_SHARD_EVAL': mode})
        mp.spawn(_cli_rank_main, args=(2, free_port(), tmp, cfg), nprocs=2, join=True)
        r0, r1 = (dict(np.load(os.path.join(tmp, 'cli%d.npz' % r))) for r in range(2))
        for k in r0:
            assert np.array_equal(r0[k], r1[k], equal_nan=True), (mode, k)
        res[mode] = r0
    # the untrained model's metrics are a pure evaluation: bit for bit; so are the step and Philox counters.  After training the two
    # RUNS differ in the last bits of their parameters (float atomics order the gradient sums inside every rank's backward: two runs
    # of the same command never agree bit for bit, sharded evaluation or not — and on 300 users one hit more or less moves recall@5 by
    # 1.7e-3), so the epochs' metrics are only required to be close.
    assert np.array_equal(res['1']['init'], res['0']['init']) and res['1']['call'] == res['0']['call'] and res['1']['t'] == res['0']['t']
    assert np.abs(res['1']['valid'] - res['0']['valid']).max() < 0.02
    assert np.isfinite(res['1']['valid']).all() and res['1']['valid'].shape[1] == 2
