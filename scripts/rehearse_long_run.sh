#!/bin/bash
# bash scripts/rehearse_long_run.sh  (one-GPU box): the driver's two-rank replicated command at FULL Electronics size and with many steps
# between flushes, gloo as the transport (both ranks on the one GPU: 2 x 48.5 GB of exposure fit) — the lazy optimizer's invariant
# (no row more than K steps behind) is checked at the final flush.  Times mean nothing.
export DCCF_DIST_BACKEND=gloo
mkdir -p gpurun_out
for steps in 300; do
port=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps $steps --warmup 30 > gpurun_out/long2.json 2> gpurun_out/long2.err
echo "steps $steps rc=$?"; tail -c 300 gpurun_out/long2.json; grep -i "error\|lazy optimizer" gpurun_out/long2.err | tail -5 | cut -c1-300
done
