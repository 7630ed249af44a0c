#!/usr/bin/env python3
"""bench.py at batch 128 for several window counts K of the lazy regularisation (DCCF_LAZY_K) with the hosted quarter of the window
on (the default) — 300 timed steps and the driver's 20-step shape.   python scripts/lazy_k_sweep.py > gpurun_out/lazy_k_sweep.json"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env, *flags):
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--cpu_baseline', '0'] + list(flags), cwd=REPO,
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    if r.returncode != 0:
        return {'error': r.stderr.decode()[-300:]}
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    k = d.get('kernel_ms', {})
    return {'ms_per_step': d['ms_per_step'], 'fwd_us': round(k.get('noise_fwd', 0) * 1e3, 2), 'bwd_us': round(k.get('k_bwd', 0) * 1e3, 2),
            'opt_us': round(k.get('opt_launch', 0) * 1e3, 2)}


out = []
for K in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['4', '6', '8', '10', '12', '16', '8'])]:
    for frac in ('0.25', '0.35'):
        env = {'DCCF_LAZY_K': str(K), 'DCCF_LAZY_HOST_FRAC': frac}
        rec = {'K': K, 'host_frac': float(frac), 'steps300': run(env, '--steps', '300', '--warmup', '30'), 'driver20': run(env, '--steps', '20', '--warmup', '5')}
        out.append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(out))
