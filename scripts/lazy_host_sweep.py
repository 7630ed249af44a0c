#!/usr/bin/env python3
"""The head of a lazy step's window hosted in the forward launch (DCCF_LAZY_HOST_FRAC): bench.py at batch
128 for several shares, 300 timed steps and the driver's 20-step shape each.  On the GPU box:
    python scripts/lazy_host_sweep.py > gpurun_out/lazy_host_sweep.json"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env, *flags):
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--cpu_baseline', '0'] + list(flags), cwd=REPO,
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    if r.returncode != 0:
        return {'error': r.stderr.decode()[-500:]}
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    k = d.get('kernel_ms', {})
    return {'ms_per_step': d['ms_per_step'], 'pairs_per_s': d['value'], 'fwd_us': round(k.get('noise_fwd', 0) * 1e3, 2),
            'bwd_us': round(k.get('k_bwd', 0) * 1e3, 2), 'opt_launch_us': round(k.get('opt_launch', 0) * 1e3, 2),
            'roofline_kernel': d['roofline'].get('kernel'), 'frac_beyond_llc': d['roofline'].get('frac_beyond_llc')}


def main():
    out = []
    fracs = [float(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0', '0.15', '0.25', '0.35', '0.5', '0'])]
    for f in fracs:
        env = {'DCCF_LAZY_HOST_FRAC': str(f)}
        rec = {'frac': f, 'steps300': run(env, '--steps', '300', '--warmup', '30'), 'driver20': run(env, '--steps', '20', '--warmup', '5')}
        out.append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
