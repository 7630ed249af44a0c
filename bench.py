# coding=utf-8
"""bench.py — DCCF training throughput on MI355X (BASELINE.json: "train pairs/sec at rank=64 Electronics").

    python bench.py [--gpus N --steps K --warmup W]        (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one optimizer step of the reference's training loop (src/runners/BaseRunner.py:175-188) over one batch of
`--batch_size` (user, positive, sampled negative) pairs: forward (Monte-Carlo exposure-weighted MF score with fresh
Gaussian feature noise), BPR loss, backward into dense-shaped gradients, and the dense l2 + clip + Adam update of EVERY
parameter.  Workload (config.workload): Electronics-shaped synthetic interactions (user_num / item_num below, Zipf item
popularity), D = 64, the 768-d feature table, dense U x I exposure matrix and both embedding tables resident in HBM;
candidates, noise, dropout masks and training negatives are drawn on the device.  Inputs are in HBM when timing starts.

The JSON line carries `roofline` (dominant kernel by measured time, HIP events on the launch stream) and `cpu_baseline`
(oracle/torch_port.py, the reference's op sequence on the host cores, bounded sample) as the task contract asks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32) dense peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=300)
    p.add_argument('--warmup', type=int, default=30)
    p.add_argument('--batch_size', type=int, default=128, help='pairs per step (reference default 128)')
    p.add_argument('--users', type=int, default=192403, help='Amazon Electronics 5-core order of magnitude')
    p.add_argument('--items', type=int, default=63001)
    p.add_argument('--dim', type=int, default=64)
    p.add_argument('--feat', type=int, default=768)
    p.add_argument('--expo', type=str, default='auto', help='dense | factors | auto (dense when it fits one GPU)')
    p.add_argument('--cpu_baseline', type=int, default=1)
    p.add_argument('--cpu_steps', type=int, default=150, help='steps of the timed CPU-baseline sample (~15 s at the best thread count)')
    p.add_argument('--seed', type=int, default=2019)
    p.add_argument('--mp', type=str, default='auto', help='multi-GPU layout: replicated (full model per GPU, one all-gather '
                                                          'per step) | sharded (BASELINE north_star: rows mod G, all-to-all row '
                                                          'exchange + all-reduce of [dW|db]) | auto (replicated while the redundant '
                                                          'dense optimizer pass is cheap: <= 1e8 params).  The layout that ran is '
                                                          'named in config.workload / config.layout')
    p.add_argument('--dp_overlap', type=int, default=1, help='replicated path: run the optimizer pass over the rows no rank '
                                                             'touches on a side stream while the gradient exchange is in flight')
    p.add_argument('--force_replicated', type=int, default=0, help='run the replicated data-parallel pipeline even at --gpus 1')
    p.add_argument('--force_sharded', type=int, default=0, help='run the row-sharded pipeline even at --gpus 1')
    p.add_argument('--prep_next', type=int, default=1, help='1: each step hands the library the next batch, whose candidates '
                   'are then drawn inside this step\'s optimizer launch (as the runner does); 0: k_prep every step')
    p.add_argument('--overlap', type=int, default=0, help='1: dccf_train_step with the untouched-row optimizer pass on a side '
                                                          'stream (see DESIGN.md: +5 %% only with DCCF_SIDE_CUS=128 --stream 1)')
    p.add_argument('--stream', type=int, default=0, help='1: run on a created stream instead of the default (null) stream')
    p.add_argument('--graph', type=int, default=0, help='1: each step is one hipGraph replay (device-side step counter); '
                   'measured slower than eager launches here (132 vs 122 us/step), hence off')
    return p.parse_args()


def synthetic_interactions(n, users, items, seed):
    """Electronics-shaped: users uniform, items Zipf(0.8) over a random permutation."""
    rng = np.random.RandomState(seed)
    w = 1.0 / np.power(np.arange(1, items + 1, dtype=np.float64), 0.8)
    w /= w.sum()
    perm = rng.permutation(items)
    uid = rng.randint(0, users, size=n).astype(np.int64)
    iid = perm[rng.choice(items, size=n, p=w)].astype(np.int64)
    key = np.unique(uid * items + iid)
    rng.shuffle(key)
    return key // items, key % items


def cpu_baseline(args, feat_cpu, seed):
    """The reference's op sequence on the host (oracle/torch_port.py), bounded: `cpu_steps` steps of the same batch
    size on the full-size embedding tables; exposure rows only for the 2000 users the sample touches.  The intra-op thread
    count is swept first (10 steps each: small ops oversubscribe a 128-thread pool) and the best one is used and reported."""
    from oracle import torch_port as TP
    us = 2000
    g = torch.Generator().manual_seed(seed)
    expo_rows = torch.randn(us, args.items, generator=g)
    model = TP.DCCFPort(args.users, args.items, args.dim, feat_cpu, expo_rows, seed=seed)
    n_sweep = 10
    batches = TP.synthetic_batches(us, args.items, args.batch_size, args.cpu_steps + 3 + 6 * (n_sweep + 1), seed=seed)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    cands = sorted({t for t in (4, 8, 16, 32, 64, 128, default_threads) if 1 <= t <= max(ncpu, default_threads)})[-6:]
    TP.train_steps(model, batches[:3])
    sweep, used = {}, 3
    for t in cands:
        torch.set_num_threads(t)
        TP.train_steps(model, batches[used:used + 1])
        t0 = time.time()
        TP.train_steps(model, batches[used + 1:used + 1 + n_sweep])
        sweep[t] = n_sweep * args.batch_size / (time.time() - t0)
        used += n_sweep + 1
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    rest = batches[used:used + args.cpu_steps]
    t0 = time.time()
    TP.train_steps(model, rest)
    dt = time.time() - t0
    torch.set_num_threads(default_threads)
    return {'value': len(rest) * args.batch_size / dt, 'unit': 'pairs/s', 'cores': best, 'kind': 'port',
            'thread_sweep_pairs_per_s': {str(k): round(v, 1) for k, v in sweep.items()}, 'host_cpus': ncpu,
            'sample': '%d steps of batch_size %d (reference op sequence in PyTorch-CPU, full-size embedding tables, '
                      'exposure rows of 2000 users) at the best of %s intra-op threads, %.1f s' % (len(rest), args.batch_size, cands, dt)}


_REAL_STDOUT = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout; RCCL prints a five-line version banner there when a communicator is created.
    Everything the process (and the libraries it loads) writes to fd 1 goes to stderr until emit() puts the line out."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.environ['DCCF_BENCH_STDOUT_FD'] = str(_REAL_STDOUT)      # (this file is also imported as module `bench`)
        os.dup2(2, 1)


def emit(obj):
    """Prints the result line on the real stdout (rank 0 calls this once)."""
    line = json.dumps(obj)
    sys.stdout.flush()
    fd = _REAL_STDOUT if _REAL_STDOUT is not None else (int(os.environ['DCCF_BENCH_STDOUT_FD'])
                                                        if 'DCCF_BENCH_STDOUT_FD' in os.environ else None)
    if fd is not None:
        os.write(fd, (line + '\n').encode())
    else:
        print(line, flush=True)


def main():
    args = parse()
    quiet_stdout()
    from dccf_amd import utils
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    # rehearsal on a box with fewer GPUs than ranks (DCCF_DIST_BACKEND=gloo): ranks share the visible devices
    backend = os.environ.get('DCCF_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(torch.cuda.device_count(), 1)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d'
                             % (args.gpus, args.gpus))
    if args.mp == 'sharded' and world == 1:
        args.force_sharded = 1              # --mp names the layout at any N (N = 1: the pipeline with a one-rank communicator)
    if args.mp == 'replicated' and world == 1:
        args.force_replicated = 1
    if world > 1:
        # one process per GPU: device binding, the GPU-count check, the process group with a finite timeout (dccf_amd/utils.py);
        # everything up to the end of the warm-up runs under a host-side deadline (a first-contact hang exits non-zero)
        rank, world = utils.init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if args.stream:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    if world > 1 or args.force_sharded or args.force_replicated:
        import datetime
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(utils.free_port()))      # (chosen by the kernel: a fixed port collides now and then)
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev,
                                    timeout=datetime.timedelta(seconds=utils.dist_timeout_s()))
        n_params = (args.users + args.items) * args.dim + args.dim * (args.dim + args.feat) + args.dim
        if args.mp == 'auto':
            args.mp = 'replicated' if n_params <= 1e8 else 'sharded'
        if (args.mp == 'sharded' or args.force_sharded) and not args.force_replicated:
            from dccf_amd import sharded
            return sharded.bench_main(args, rank, world, dev)
        from dccf_amd import replicated
        return replicated.bench_main(args, rank, world, dev)

    from dccf_amd import _lib as L
    from dccf_amd.models import DCCF, FusedOptimizer
    from dccf_amd.data_processor import DeviceTrainSet

    U, I, D, F, B = args.users, args.items, args.dim, args.feat, args.batch_size
    S, A, std, p_drop, lr, l2 = 10, 2, 0.1, 0.2, 1e-3, 1e-4
    g = torch.Generator(device=dev).manual_seed(args.seed)
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
    model = DCCF(path=None, dataset=None, sentence_model=None, sample_num=S, attribute_num=A, std=std, label_min=0,
                 label_max=1, feature_num=0, user_num=U, item_num=I, u_vector_size=D, i_vector_size=D, n_layers=1,
                 random_seed=args.seed, model_path='/tmp/bench.pt', feature_embedding=feat, expo_prob=expo, ips_factors=ips)
    model.apply(model.init_paras)
    opt = FusedOptimizer(model, 'adam', lr, l2)
    model.optimizer = opt
    model.train()

    n_pairs = (args.steps + args.warmup + 2) * B
    uid, iid = synthetic_interactions(int(n_pairs * 1.15) + 1000, U, I, args.seed)
    uid, iid = uid[:n_pairs], iid[:n_pairs]
    ds = DeviceTrainSet(uid, iid, U, I, args.seed)
    model.ctx.reserve(2 * B, D, F, S, A)
    y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
    batch = {'Y': y, 'rank': 1, 'train': True, 'dropout': p_drop}

    def run(full, k0, k1):      # one library call per step (dccf_train_step)
        for k in range(k0, k1):
            batch['X'] = full[k]
            model.train_step(batch, overlap=args.overlap, X_next=full[k + 1] if args.prep_next and k + 1 < k1 else None)

    from dccf_amd.models import StepGraph
    sg = StepGraph(model, opt, args.warmup + args.steps, 2 * B, p_drop) if args.graph else None

    def epoch_tensor(e):
        full, _ = ds.epoch_batches(e, B)
        return full[:args.warmup + args.steps].contiguous()

    if sg is not None:
        sg.load_epoch(epoch_tensor(0))
        for _ in range(args.warmup):
            sg.step()
    else:
        full = epoch_tensor(0)
        run(full, 0, args.warmup)
    # the warm-up ends like the timed region does — with the flush of the rows the lazy regularisation left behind — so that every
    # kernel of the timed region (the flush's instances too) has run once before the clock starts
    opt.flush()
    torch.cuda.synchronize()

    t0 = time.perf_counter()
    if sg is not None:
        # timed: the epoch's negative sampling + batch tensor + K graph replays.  (The warm-up consumed `warmup` slots of
        # the ring, load_epoch places the new epoch's batches where the next steps look.)
        sg.load_epoch(epoch_tensor(1))
        for _ in range(args.steps):
            sg.step()
    else:
        full = epoch_tensor(1)
        run(full, args.warmup, args.warmup + args.steps)
    # the lazy regularisation leaves rows up to K steps behind: bringing them up to date (what evaluation / a checkpoint would
    # trigger) belongs to the timed work — every parameter has received every one of the K steps when the clock stops
    opt.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # ---- per-kernel durations, measured live with HIP events on the launch stream; none of this is in the timed region.
    # Pass A = the launch structure the timed region runs (dccf_train_step with X_next: forward, backward + hosted slice of the
    # optimizer pass, optimizer launch + the next step's preparation), bracketed inside the library (dccf_profile).
    # Pass B = the split calls (forward/backward, then the WHOLE dense pass as one launch), for the stand-alone pass.
    n_prof = min(args.steps, 100)
    full = epoch_tensor(2)
    # ~25 ms of queued streaming work first: the host enqueues the whole pass while the GPU is still busy, so no bracket
    # contains a host launch gap (on a slow host the brackets of an idle queue read 5-10 us long)
    blocker = torch.zeros(256 << 20, device=dev)

    def fill_queue():
        for _ in range(64):
            blocker.add_(1.0)

    def profile_structure(full):
        """n_prof steps of the timed launch structure with the library's per-launch event brackets; returns (ms per launch by
        kernel with the event-boundary cost subtracted, launch counts, that cost, item rows hosted in the backward launch)."""
        # [event, event] with nothing between: what one event boundary costs on this stream — every bracket contains it once
        empties = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(n_prof)]
        model.ctx.profile(True)
        fill_queue()
        for k in range(n_prof):
            batch['X'] = full[k]
            model.train_step(batch, overlap=args.overlap, X_next=full[k + 1] if args.prep_next and k + 1 < n_prof else None)
            empties[k][0].record()
            empties[k][1].record()
        torch.cuda.synchronize()
        prof = model.ctx.profile_read()
        model.ctx.profile(False)
        ev = sum(a.elapsed_time(b) for a, b in empties) / n_prof
        return ({k: max(v[0] / max(v[1], 1) - ev, 1e-6) for k, v in prof.items()}, {k: int(v[1]) for k, v in prof.items()}, ev,
                model.ctx.hosted_rows())

    timed, timed_counts, ev_ms, hosted_rows = profile_structure(full)
    lazy = opt.lazy
    lazy_rows = int((lazy.step_list(opt.t, 2 * B * (S + 2)) >= 0).sum()) if lazy is not None else 0       # distinct rows of the last step
    dense_timed = None
    if lazy is not None:
        # the same steps with the dense pass every launch (lazy_K = 0): the HBM-bound form of the optimizer launch, for the record
        model.lazy_K = 0
        model._opt_struct_key = None          # (the next train_step flushes the lazy rows and rebuilds its argument block)
        dense_timed, _, _, dense_hosted = profile_structure(epoch_tensor(4))
        model.lazy_K, model._opt_struct_key = lazy.K, None
    model.ctx.profile(True)
    events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(n_prof)]
    full = epoch_tensor(3)
    fill_queue()
    for k in range(n_prof):
        batch['X'] = full[k]
        model(batch)
        events[k][0].record()
        opt.step()
        events[k][1].record()
    torch.cuda.synchronize()
    split = {k: max(v[0] / max(v[1], 1) - ev_ms, 1e-6) for k, v in model.ctx.profile_read().items()}
    model.ctx.profile(False)
    whole_pass_ms = max(sum(a.elapsed_time(b) for a, b in events) / n_prof - ev_ms, 1e-6)
    del blocker
    n_params = model.flat_p.numel()
    L_rows = 2 * B * (S + 1) * A
    # ---- the same optimizer kernel on a working set far beyond the 256 MiB Infinity Cache (p + m + v = 3.2 GB): at
    # Electronics size p + m + v = 197 MB sits in that cache and FETCH_SIZE counts its hits, so the in-step figure is a
    # cache-assisted one; this is the HBM figure
    big_rows = 4 << 20
    bp, bg = torch.zeros(big_rows * 64, device=dev), torch.zeros(big_rows * 64, device=dev)
    b1, b2 = torch.zeros_like(bp), torch.zeros_like(bp)
    bfl = torch.zeros(big_rows, dtype=torch.uint8, device=dev)
    bp.normal_(0.0, 0.01)
    big_ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(12)]
    for i in range(12):
        big_ev[i][0].record()
        L.dense_opt_step_rows('adam', bp, bg, b1, b2, lr, l2, l2, 50.0, i + 1, [(0, big_rows, 64, bfl)])
        big_ev[i][1].record()
    torch.cuda.synchronize()
    big_ms = sum(a.elapsed_time(b) for a, b in big_ev[2:]) / 10 - ev_ms
    big_gb = 24.0 * big_rows * 64 / 1e9
    del bp, bg, b1, b2, bfl
    # ---- algorithmic work per launch.  Dense Adam: p, m, v read + written = 24 B per parameter
    # (the gradient is touched only for the rows the step reached; SURVEY.md section 8d prices a gradient-streaming Adam at 28).
    fwd_tf = 2.0 * L_rows * (D + F) * D / 1e12
    bwd_tf = (2.0 * L_rows * (D + F) * D + 2.0 * L_rows * D * D) / 1e12
    kernels = {'opt_launch': timed.get('opt_launch', whole_pass_ms), 'noise_fwd': timed['noise_fwd'], 'k_bwd': timed['noise_bwd_eps']}
    if 'lazy_catchup' in timed:
        kernels['k_lazy_catchup (unprepared steps only: %d of %d; otherwise a role of the previous optimizer launch)'
                % (timed_counts['lazy_catchup'], n_prof)] = timed['lazy_catchup']
    if 'prep' in timed:
        kernels['k_prep (unprepared steps only: %d of %d)' % (timed_counts['prep'], n_prof)] = timed['prep']
    pmc, pmc_all = None, {}
    try:
        pmc_all = json.load(open(os.path.join(REPO, 'profiles', 'r03_pmc_traffic.json')))
        pmc = pmc_all['_meta']
    except Exception:
        pass
    PMC_NOTE = ('MB of HBM traffic per launch; NOT measured in this run: the separate rocprofv3 --pmc FETCH_SIZE (doubled, as '
                'MI355X_MICROARCH.md prescribes for gfx950) / --pmc WRITE_SIZE passes of the same command at this batch size, committed as '
                'profiles/r03_pmc_traffic.json')

    def pmc_mb(short):
        v = pmc_all.get(short)
        return round(v['total_mb'], 2) if (v and B == 128 and D == 64 and F == 768) else None

    def dense_launch(ms, hosted):
        """The HBM-bound form of the optimizer launch: 24 B per parameter it streams."""
        gb = 24.0 * (n_params - hosted * D) / 1e9
        return {'kernel': 'k_dense_opt_rows<Adam> as launched by dccf_train_step (U, W, b, the marked rows of V and the rows of V '
                          'not hosted in the backward launch; + the next step\'s preparation)',
                'bound': 'hbm', 'achieved': round(gb / (ms / 1e3), 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(gb / (ms / 1e3) / HBM_PEAK_GBS, 4),
                'traffic': round(pmc['dense_adam_bytes_per_param'] * (n_params - hosted * D) / 1e9, 4) if pmc else None,
                'traffic_unit': 'GB per launch; NOT measured in this run: bytes/param of the separate rocprofv3 --pmc FETCH_SIZE / '
                                'WRITE_SIZE passes committed as profiles/r03_pmc_traffic.json',
                'algorithmic_per_launch': round(gb, 4), 'avg_launch_ms': round(ms, 5),
                'hosted_in_backward_launch': {'item_rows': hosted, 'GB': round(24.0 * hosted * D / 1e9, 4)}}

    dom = max(('opt_launch', 'noise_fwd', 'k_bwd'), key=lambda k: kernels[k])
    if dom == 'opt_launch' and lazy is None:
        roofline = dense_launch(kernels[dom], hosted_rows)
    elif dom == 'opt_launch':
        # windowed lazy regularisation: the launch streams 1 / K of the row-structured parameters (24 B each), the rows the step
        # touched (32 B: + gradient read and re-zeroed) and the dense tail W, b (32 B) — and replays up to K steps per element
        n_rowp = (U + I) * D
        gb = (24.0 * n_rowp / lazy.K + 32.0 * lazy_rows * D + 32.0 * (n_params - n_rowp)) / 1e9
        roofline = {'kernel': 'k_lazy_opt<Adam> (windowed lazy regularisation, K = %d: the rows the step touched, W, b and one K-th of '
                              'the other rows, advanced K steps in registers; + the next step\'s preparation)' % lazy.K,
                    'bound': 'hbm', 'achieved': round(gb / (kernels[dom] / 1e3), 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(gb / (kernels[dom] / 1e3) / HBM_PEAK_GBS, 4), 'traffic': pmc_mb('lazy_opt (K = 8)'),
                    'traffic_unit': PMC_NOTE,
                    'algorithmic_per_launch': round(gb, 4), 'avg_launch_ms': round(kernels[dom], 5),
                    'note': 'NOT an HBM-bound launch any more: the replay (2 IEEE divisions + 1 square root per element and step, the '
                            'dense pass\'s exact operation order) is bound by the vector ALU — 16.4 M element-steps cost >= 21 us '
                            'whatever K is (profiles/r02_lazy_replay_bench.md); the same step with the dense pass is in dense_mode',
                    'element_steps_per_launch': n_rowp}
    else:
        tf = fwd_tf if dom == 'noise_fwd' else bwd_tf
        achieved = tf / (kernels[dom] / 1e3)
        roofline = {'kernel': dom, 'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                    'traffic': pmc_mb('noise_fwd' if dom == 'noise_fwd' else 'noise_bwd_eps'), 'traffic_unit': PMC_NOTE,
                    'algorithmic_per_launch': round(tf, 6), 'avg_launch_ms': round(kernels[dom], 5)}
    roofline.update({
        'launch_structure': 'dccf_train_step with X_next: the launches the timed region runs, bracketed by HIP events inside the '
                            'library on the launch stream; event boundary %.4f ms subtracted' % ev_ms,
        # the whole dense pass as ONE launch (split calls): same kernel, nothing hosted
        'whole_pass': {'avg_launch_ms': round(whole_pass_ms, 5), 'algorithmic_GB': round(24.0 * n_params / 1e9, 4),
                       'achieved': round(24.0 * n_params / 1e9 / (whole_pass_ms / 1e3), 1),
                       'frac': round(24.0 * n_params / 1e9 / (whole_pass_ms / 1e3) / HBM_PEAK_GBS, 4)},
        # p + m + v of this model (197 MB) fit the 256 MiB Infinity Cache: the dense figures are cache-assisted.  The same
        # kernel on 3.2 GB of state:
        'frac_beyond_llc': round(big_gb / (big_ms / 1e3) / HBM_PEAK_GBS, 4),
        'beyond_llc': {'params': big_rows * 64, 'algorithmic_GB': round(big_gb, 3), 'avg_launch_ms': round(big_ms, 4),
                       'achieved': round(big_gb / (big_ms / 1e3), 1)},
        'event_boundary_ms': round(ev_ms, 5)})
    if dense_timed is not None:
        d_ms = sum(dense_timed.get(k, 0.0) for k in ('noise_fwd', 'noise_bwd_eps', 'opt_launch'))
        roofline['dense_mode'] = dict(dense_launch(dense_timed['opt_launch'], dense_hosted),
                                      step_kernel_ms={'noise_fwd': round(dense_timed['noise_fwd'], 5),
                                                      'k_bwd': round(dense_timed['noise_bwd_eps'], 5),
                                                      'opt_launch': round(dense_timed['opt_launch'], 5), 'sum': round(d_ms, 5)},
                                      how='DCCF_LAZY_K=0 (or model.lazy_K = 0): every optimizer launch streams every parameter')
    roofline_mfma = {
        'peak_TFLOPs': MFMA_F32_PEAK_TFLOPS,
        'noise_fwd': {'flops': fwd_tf * 1e12, 'avg_us': round(kernels['noise_fwd'] * 1e3, 2),
                      'TFLOPs': round(fwd_tf / (kernels['noise_fwd'] / 1e3), 2),
                      'frac': round(fwd_tf / (kernels['noise_fwd'] / 1e3) / MFMA_F32_PEAK_TFLOPS, 4)},
        'k_bwd': {'flops': bwd_tf * 1e12, 'avg_us': round(kernels['k_bwd'] * 1e3, 2),
                  'TFLOPs': round(bwd_tf / (kernels['k_bwd'] / 1e3), 2),
                  'frac': round(bwd_tf / (kernels['k_bwd'] / 1e3) / MFMA_F32_PEAK_TFLOPS, 4),
                  'note': 'the launch also hosts %.1f MB of the optimizer pass' % (24.0 * hosted_rows * D / 1e6) if hosted_rows else
                          'nothing hosted'}}
    kernels['dense_adam_whole_pass (split calls)'] = whole_pass_ms
    value = args.steps * B / dt
    out = {
        'metric': 'train pairs/sec at rank=64 Electronics', 'value': round(value, 1), 'unit': 'pairs/s', 'n_gpus': 1,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'DCCF train step (fwd + BPR + bwd + dense l2/clip/Adam), Electronics-shaped synthetic: '
                               'user_num=%d item_num=%d D=%d F=%d S=%d A=%d, exposure=%s, fused on-device negatives'
                               % (U, I, D, F, S, A, expo_mode),
                   'batch_size': B, 'optimizer': 'Adam lr=1e-3 l2=1e-4 dropout=0.2', 'params': n_params,
                   'regularisation': ('windowed lazy (K = %d): untouched rows are advanced K steps at a time, bit-identical to the '
                                      'dense pass' % lazy.K) if lazy is not None else 'dense pass every step',
                   'step_launch': 'hipGraph replay' if args.graph else 'eager launches'},
        'roofline': roofline,
        'roofline_mfma': roofline_mfma,
        'kernel_ms': {k: round(v, 5) for k, v in sorted(kernels.items())},
        'embedding_fwd_bwd': {   # 4D(3+6(S+1)) + 8F + 8(S+1) + 32 algorithmic bytes per pair
            'bytes_per_pair': 4 * D * (3 + 6 * (S + 1)) + 8 * F + 8 * (S + 1) + 32,
            'kernel_ms': round(split['noise_fwd'] + split['noise_bwd_eps'] + split.get('pair_epilogue', 0.0), 5)},
    }
    ebp = out['embedding_fwd_bwd']
    ebp['achieved_GBps'] = round(B * ebp['bytes_per_pair'] / (ebp['kernel_ms'] / 1e3) / 1e9, 2)
    if args.cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args, feat.cpu(), args.seed)
    emit(out)


if __name__ == '__main__':
    main()
