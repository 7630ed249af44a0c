#!/usr/bin/env python3
"""SQ counters of the two MFMA kernels (k_noise_fwd, k_bwd): how the wave cycles split into issuing, issue stalls and waits, how
busy the matrix pipe is, waves launched, LDS bank conflicts.  Separate rocprofv3 --pmc passes (no trace domains beside them).
Run on the GPU box from the repo root:  python scripts/mfma_pmc.py [batch_size] [out.md]  ->  gpurun_out/<out.md> (copy it
into profiles/; round 1: B = 4096 -> profiles/r01_mfma_pmc.md, round 3: B = 128 -> profiles/r03_mfma_pmc_b128.md)"""
import csv
import glob
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = [['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'],
        ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU'],
        ['SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_COEXEC_CYCLES'],
        ['SQ_WAVES', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'],
        ['SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR'],
        ['GRBM_GUI_ACTIVE', 'SQ_INST_CYCLES_VMEM', 'SQ_WAIT_INST_LDS']]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
OUT_MD = sys.argv[2] if len(sys.argv) > 2 else 'mfma_pmc_b%d.md' % B


def main():
    res = {}
    for i, cs in enumerate(SETS):
        out = os.path.join(REPO, 'gpurun_out', 'mfma_pmc_%d' % i)
        cmd = ['rocprofv3', '--pmc'] + cs + ['--output-format', 'csv', '-d', out, '--', 'python3', os.path.join(REPO, 'bench.py'),
                                             '--batch_size', str(B), '--steps', '100', '--warmup', '10', '--cpu_baseline', '0']
        r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        fs = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
        if r.returncode != 0 or not fs:
            print('pass %d failed: %s' % (i, r.stderr.decode()[-400:]), file=sys.stderr)
            continue
        for row in csv.DictReader(open(fs[0])):
            k = 'k_noise_fwd' if 'k_noise_fwd' in row['Kernel_Name'] else ('k_bwd' if 'k_bwd' in row['Kernel_Name'] else None)
            if k:
                a = res.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, 0])
                a[0] += float(row['Counter_Value'])
                a[1] += 1
    lines = ['# rocprofv3 --pmc (separate passes), `python3 bench.py --batch_size %d --steps 100 --warmup 10 --cpu_baseline 0`' % B, '',
             'Per launch, summed over the chip.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave;',
             'WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.', '',
             '| kernel | counter | per launch | share of SQ_WAVE_CYCLES |', '|---|---|---:|---:|']
    for k, cs in res.items():
        wc = cs.get('SQ_WAVE_CYCLES', [0, 1])
        wc = wc[0] / max(wc[1], 1)
        for c, (v, n) in cs.items():
            v = v / max(n, 1)
            share = ('%.1f %%' % (100 * v / wc)) if wc and c.startswith(('SQ_WAIT', 'SQ_ACTIVE_INST')) else ''
            lines.append('| `%s` | %s | %.4g | %s |' % (k, c, v, share))
    open(os.path.join(REPO, 'gpurun_out', OUT_MD), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
