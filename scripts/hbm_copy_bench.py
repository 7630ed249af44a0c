"""HBM copy / triad micro-benchmark (torch ops): what a plain streaming kernel reaches on this box, for the 'of measured copy' figure."""
import torch, time, json
dev = torch.device('cuda', 0)
out = {}
for mb in (64, 256, 1024, 4096):
    n = mb * (1 << 20) // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a); c = torch.randn(n, device=dev)
    for name, fn, nbytes in (('copy', lambda: b.copy_(a), 8 * n), ('add', lambda: torch.add(a, c, out=b), 12 * n),
                             ('scale_inplace', lambda: a.mul_(1.0001), 8 * n)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out['%s_%dMB' % (name, mb)] = round(nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12, 3)
print(json.dumps(out))
