#!/bin/bash
# the driver's multi-GPU launch line, at one process (all that a 1-GPU box can run)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
