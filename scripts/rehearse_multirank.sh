#!/bin/bash
# bash scripts/rehearse_multirank.sh  (on the one-GPU box): the driver's multi-rank commands with the ranks sharing the device and
# gloo as the transport (DCCF_DIST_BACKEND=gloo) — launch, schedule, collectives' call pattern, cross-checks, deadlines and the JSON
# line end to end.  The TIMES of such a run mean nothing (gloo stages every buffer through the host) and are not recorded.
OUT=$PWD/gpurun_out/rehearsal
mkdir -p $OUT
export DCCF_DIST_BACKEND=gloo
run() {   # name, nproc, extra flags
  local name=$1 n=$2; shift 2
  local port=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $n --steps 20 --warmup 5 --users 30000 --items 9000 "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); c=d['config']
    print('n_gpus',d['n_gpus'],'layout',c.get('layout'),'collectives',c.get('collectives'),'replicas_bit_identical',c.get('replicas_bit_identical'),'agreed_after_warmup',c.get('replicas_agreed_after_warmup'))
except Exception as e: print('NO LINE',e)
")"
}
run repl2 2 --mp replicated
run repl4 4 --mp replicated
run shard2 2 --mp sharded
run shard4 4 --mp sharded
# a rank that never joins: the others must exit non-zero with the deadline's message instead of hanging
port=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
( RANK=0 LOCAL_RANK=0 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port DCCF_DIST_TIMEOUT_S=20 DCCF_WARMUP_DEADLINE_S=25 \
  timeout -k 10 120 python bench.py --gpus 2 --steps 20 --warmup 5 --users 30000 --items 9000 > $OUT/lonely.json 2> $OUT/lonely.err; echo "lonely rank rc=$? (expected non-zero)"; tail -2 $OUT/lonely.err | cut -c1-300 )
