#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multirank_one_gpu.py -m gpu -q -x > gpurun_out/multirank.log 2>&1
tail -40 gpurun_out/multirank.log
