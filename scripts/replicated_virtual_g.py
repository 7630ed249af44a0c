"""python scripts/replicated_virtual_g.py  (VG=<ranks>)  — dev tool: the replicated step's KERNEL work at G-rank shape on ONE GPU.  Rank 0 of a virtual world of G: the schedule has
G batches per step, every rank's rows are marked / claimed / imported as at world size G, but the all-gather is replaced by
a local copy of rank 0's buffer into the G slots (values of the other ranks' rows are therefore rank 0's gradients: finite,
meaningless — this measures launches, not results, and says nothing about xGMI)."""
import os, sys, time, torch
sys.path.insert(0, '/root/repo'); sys.argv = ['bench.py']
import bench
from dccf_amd import replicated
from dccf_amd.data_processor import DeviceTrainSet
dev = torch.device('cuda:0')
U, I, D, F, B, S, A = 192403, 63001, 64, 768, 128, 10, 2
G = int(os.environ.get('VG', '8'))
steps, warm = int(os.environ.get("VSTEPS", "200")), 30
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.randn(I, F, generator=g, device=dev) * 0.05
ips = dict(P=torch.randn(U, 64, generator=g, device=dev) * 0.1, Q=torch.randn(I, 64, generator=g, device=dev) * 0.1,
           bu=torch.randn(U, generator=g, device=dev) * 0.1, bi=torch.randn(I, generator=g, device=dev) * 0.1,
           prop=torch.rand(I, generator=g, device=dev), b0=0.1, M=0.1)
be = replicated.HipBackend(dev)
tr = replicated.ReplicatedDCCF(0, G, U, I, D, S, A, 0.1, 0.2, 1e-3, 1e-4, 1, be, dev, feat, ips=ips, max_rows=2 * B, overlap=True)
tr.init_params()
n_pairs = (steps + warm + 2) * B * G
uid, iid = bench.synthetic_interactions(int(n_pairs * 1.15) + 1000, U, I, 1)
ds = DeviceTrainSet(uid[:n_pairs], iid[:n_pairs], U, I, 1)
full, _ = ds.epoch_batches(0, B)
sched = full[:(steps + warm) * G].view(steps + warm, G, 2 * B, 2)
y = torch.cat([torch.ones(B, device=dev), torch.zeros(B, device=dev)])
pred = torch.empty(2 * B, device=dev)

def step(k, last):
    X_all = sched[k]
    X_all_next = None if last else sched[k + 1]
    tr.next = None if X_all_next is None else (X_all_next[0], X_all_next)
    step0, t = tr.t * tr.G, tr.t + 1
    be.local(tr, X_all[0], y, step0, pred, X_all, step0)
    tr.bufs.view(G, -1).copy_(tr.buf)            # the stand-in for the all-gather
    tr.t = t
    be.overlap(tr, t)
    be.finish(tr, t, True)
    tr.parity ^= 1

for k in range(warm):
    step(k, False)      # (every step announces the next: in an UNannounced step the ids travel in the buffers, and this stand-in
                        # copies rank 0's ids into every slot — the other virtual ranks' rows would be claimed and never updated,
                        # which the lazy optimizer's invariant check reports at the flush)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(warm, warm + steps):
    step(k, k == warm + steps - 1)
tr.flush()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('virtual G = %d: %.4f ms/step (one GPU, gather = local copy), prepared steps %d' % (G, dt / steps * 1e3, be.ctx.prepared_steps()), flush=True)
