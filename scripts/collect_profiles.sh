#!/bin/bash
# bash scripts/collect_profiles.sh  (on the GPU box, from the repo root): the bench lines, rocprofv3 kernel statistics of the three
# layouts and the PMC traffic passes behind profiles/r03_*; everything lands in gpurun_out/final/ (copy what is to be kept).
set -e
OUT=$PWD/gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $OUT/bench_lazy.json 2> $OUT/bench_lazy.err
echo "bench lazy done"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu_baseline 0 > $OUT/bench_lazy_driver.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu_baseline 0 > $OUT/bench_lazy_driver2.json 2>/dev/null
DCCF_LAZY_K=0 timeout -k 10 300 python bench.py --cpu_baseline 0 > $OUT/bench_dense.json 2>/dev/null
DCCF_LAZY_K=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu_baseline 0 > $OUT/bench_dense_driver.json 2>/dev/null
echo "bench dense done"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lazy -- python3 $OLDPWD/bench.py --cpu_baseline 0 > /dev/null 2>&1
export DCCF_LAZY_K=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dense -- python3 $OLDPWD/bench.py --cpu_baseline 0 > /dev/null 2>&1
unset DCCF_LAZY_K
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_repl -- python3 $OLDPWD/bench.py --mp replicated --cpu_baseline 0 > $OUT/bench_replicated.json 2>/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_shard -- python3 $OLDPWD/bench.py --mp sharded --cpu_baseline 0 > $OUT/bench_sharded.json 2>/dev/null
cd $OLDPWD
echo "rocprof done"
for d in lazy dense repl shard; do f=$(find $OUT/prof_$d -name "*kernel_stats.csv" | head -1); python scripts/prof_summary.py $f > $OUT/stats_$d.md; rm -rf $OUT/prof_$d; done
PMC_OUT=gpurun_out/final timeout -k 10 600 python scripts/pmc_traffic.py r03 > $OUT/pmc.log 2>&1
rm -rf gpurun_out/pmc
echo "all done"
