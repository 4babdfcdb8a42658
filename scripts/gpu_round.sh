#!/bin/bash
# round 3, batch 20: two-stream right-sided solve sweep (predict variance / conditional covariance), lower-only covariance product
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b20
mkdir -p $O
export TMPDIR=/tmp
{
date
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -x -q -k "not full_size and not stress" 2>&1 | tail -4
timeout 300 python scripts/time_paths.py 16384 4096
TGP_HIP_OPTIONS=lookahead=0 timeout 300 python scripts/time_paths.py 16384 4096 | grep -i "variance\|covariance"
timeout 300 python scripts/time_paths.py 4096 1024 | grep -i "variance\|covariance"
timeout 300 python scripts/time_paths.py 65536 4096 | grep -i "variance\|covariance\|predict"
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-300
