# coding=utf-8
"""Prints VGPRs / spills / scratch per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py dccf_amd/csrc/dccf_kernels.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-ffp-contract=off',
                      '-Wno-unused-result', '-c', src, '-o', '/tmp/_kr.o', '-Rpass-analysis=kernel-resource-usage'],
                     stderr=subprocess.PIPE, stdout=subprocess.PIPE, universal_newlines=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+(\w[\w ]*?)(?: \[[\w/]+\])?: (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k in sorted(rows):
    name = subprocess.run(['c++filt', k], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
    name = re.sub(r'\(.*', '', name)
    if flt in name:
        r = rows[k]
        print('%-60s VGPR %3d  spill %3d  scratch %4d  occ %s' % (name, r.get('VGPRs', -1), r.get('VGPRs Spill', -1),
                                                                 r.get('ScratchSize', -1), r.get('Occupancy', '?')))
