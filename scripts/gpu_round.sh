#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TGP_BENCH_ONE_GPU=1
date
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 1 --warmup 1 > gpurun_out/rehearsal_default.json 2> gpurun_out/rehearsal_default.err
echo rc=$?
date
tail -1 gpurun_out/rehearsal_default.json | cut -c1-1800
tail -5 gpurun_out/rehearsal_default.err
