#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE in SEPARATE --pmc passes (no trace domains beside them), FETCH_SIZE doubled for wide coalesced streaming reads on
gfx950.  Run on the GPU box from the repo root:

    python scripts/pmc_traffic.py [r02]      # -> profiles/<round>_pmc_traffic.{md,json}

The profiled command is `python3 bench.py --steps 100 --warmup 10 --cpu_baseline 0` (the program itself after `--`)."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {'k_prep': 'prep', 'k_noise_fwd': 'noise_fwd', 'k_pair_epilogue': 'pair_epilogue', 'k_bwd': 'noise_bwd_eps',
         'k_dense_opt_rows': 'dense_adam', 'k_dense_opt_rows_beyond_llc': 'dense_adam_beyond_llc (268 M parameters)',
         'k_lazy_opt': 'lazy_opt (K = 8)', 'k_lazy_catchup': 'lazy_catchup', 'k_lazy_mark': 'lazy_mark'}


N_PARAMS = None
ROUND = sys.argv[1] if len(sys.argv) > 1 else 'r02'
# (gpurun merges only gpurun_out/ back: write there on the GPU box, then copy the two files into profiles/)
OUTDIR = os.path.join(REPO, os.environ.get('PMC_OUT', 'profiles'))


def collect(counter, outdir):
    # DCCF_NO_HOSTV: every optimizer launch is then the ordinary whole pass — the launch bench.py's roofline describes
    env = dict(os.environ, TMPDIR='/tmp', DCCF_NO_HOSTV='1')
    cmd = ['rocprofv3', '--pmc', counter, '--output-format', 'csv', '-d', outdir, '--', 'python3', os.path.join(REPO, 'bench.py'),
           '--steps', '100', '--warmup', '10', '--cpu_baseline', '0']
    r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    global N_PARAMS
    for line in r.stdout.decode().splitlines():
        if line.startswith('{') and '"params"' in line:
            N_PARAMS = json.loads(line)['config']['params']
    f = glob.glob(os.path.join(outdir, '**', '*counter_collection.csv'), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        full = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.match(r'(?:void )?([\w:]+)', full).group(1)
        # bench.py also runs the optimizer kernel on a 3.2 GB buffer (roofline.frac_beyond_llc): a one-segment instance, kept apart
        # (since round 3: the non-temporal, one-slot-per-thread instance <KIND, 1, false, true>; round 2: <KIND, 2, ...>)
        # (template arguments <KIND, UN, TO, NT[, GW]>)
        if name == 'k_dense_opt_rows' and re.search(r'k_dense_opt_rows<\d+, (2,|\d+, (false|true|0|1), (true|1)[,>])', full):
            name = 'k_dense_opt_rows_beyond_llc'
        a = acc.setdefault(name, [0.0, 0])
        a[0] += float(r['Counter_Value'])
        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    out = os.path.join(REPO, 'gpurun_out', 'pmc')
    fetch = collect('FETCH_SIZE', os.path.join(out, 'fetch'))
    write = collect('WRITE_SIZE', os.path.join(out, 'write'))
    res, lines = {}, []
    for k, short in NAMES.items():
        if k not in fetch or k not in write:
            continue
        fk, n = fetch[k]
        wk = write[k][0]
        rd, wr = 2.0 * fk * 1024 / 1e6, wk * 1024 / 1e6
        res[short] = {'fetch_kb': fk, 'write_kb': wk, 'read_mb_corrected': rd, 'write_mb': wr, 'total_mb': rd + wr}
        lines.append('| `%s` | %d | %.1f | %.2f | %.1f | %.2f | %.2f |' % (k, n, fk, rd, wk, wr, rd + wr))
    n_params = N_PARAMS
    if n_params and 'dense_adam' in res:
        res['_meta'] = {'params': n_params, 'dense_adam_bytes_per_param': res['dense_adam']['total_mb'] * 1e6 / n_params}
    with open(os.path.join(OUTDIR, ROUND + '_pmc_traffic.json'), 'w') as f:
        json.dump(res, f, indent=1)
    md = ['# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), `DCCF_NO_HOSTV=1 python3 bench.py --steps 100 --warmup 10 --cpu_baseline 0` (every optimizer launch is then the whole pass)',
          '', 'FETCH_SIZE is in KB and on gfx950 reports exactly half of a wide coalesced streaming read (MI355X_MICROARCH.md §HBM) — the',
          '`read MB` column doubles it; that correction is calibrated for 16-B-per-lane streams (the dense optimizer); the gather',
          "kernels' reads are uncalibrated.  Regenerate with `python scripts/pmc_traffic.py` on the GPU box.", '',
          '| kernel | launches | FETCH_SIZE KB | read MB (x2) | WRITE_SIZE KB | write MB | total MB / launch |', '|---|---:|---:|---:|---:|---:|---:|'] + lines
    if '_meta' in res:
        d = res['dense_adam']
        md += ['', 'Row-aware dense Adam (`k_dense_opt_rows`, %d parameters incl. padding): %.2f MB read = %.2f B/param, %.2f MB written = '
               '%.2f B/param -> %.1f MB of HBM traffic per launch against 24 B/param = %.1f MB algorithmic (p, m, v read + written; the '
               'gradient is read and re-zeroed only for the ~3 k rows a step touches).'
               % (n_params, d['read_mb_corrected'], d['read_mb_corrected'] * 1e6 / n_params, d['write_mb'], d['write_mb'] * 1e6 / n_params,
                  d['total_mb'], 24.0 * n_params / 1e6)]
    open(os.path.join(OUTDIR, ROUND + '_pmc_traffic.md'), 'w').write('\n'.join(md) + '\n')
    print('\n'.join(md))


if __name__ == '__main__':
    main()
