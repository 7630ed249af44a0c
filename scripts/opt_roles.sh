#!/bin/bash
# which role of the optimizer launch is the long pole once the window is not? (window share hosted in the forward: 0.6)
run() { echo "== $1"; env $1 python bench.py --cpu_baseline 0 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('ms_per_step',d['ms_per_step'],'fwd',round(k['noise_fwd']*1e3,2),'bwd',round(k['k_bwd']*1e3,2),'opt',round(k['opt_launch']*1e3,2))"; }
run "DCCF_LAZY_HOST_FRAC=0.6"
run "DCCF_LAZY_HOST_FRAC=0.6 DCCF_LAZY_NO_CU=1"
run "DCCF_LAZY_HOST_FRAC=0.6 DCCF_NO_GW_PART=1"
run "DCCF_LAZY_HOST_FRAC=0.6 DCCF_LAZY_NO_CU=1 DCCF_NO_GW_PART=1"
run "DCCF_LAZY_HOST_FRAC=0.25"
run "DCCF_LAZY_HOST_FRAC=0.25 DCCF_NO_GW_PART=1"
run "DCCF_LAZY_HOST_FRAC=0.25 DCCF_LAZY_NO_CU=1"
