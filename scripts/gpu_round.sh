#!/bin/bash
# round 2, batch 8: helper-workgroup split of the streaming solves, Kernel.matmul in one pass
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
{
echo "== pytest -m gpu (stream_trsv=1 auto)"; date
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
echo "== pytest solves with the split forced"; date
TGP_HIP_OPTIONS=stream_trsv=2 timeout 900 python -m pytest tests/test_gpu_gp.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4
for v in 3 2 1; do
echo "== adjacent paths, stream_trsv=$v"; date
TGP_HIP_OPTIONS=stream_trsv=$v timeout 200 python scripts/time_paths.py 16384 4096 | grep -E "resident|predict mean at"
TGP_HIP_OPTIONS=stream_trsv=$v timeout 200 python scripts/time_paths.py 65536 4096 2>&1 | grep -E "resident|predict mean at"
done
echo "== kernel times"; date
cd /tmp
for v in 3 2; do
TGP_HIP_OPTIONS=stream_trsv=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solves_$v -o bench -- python $R/scripts/time_paths.py 16384 4096 > /dev/null 2>&1
echo "-- stream_trsv=$v"; python $R/scripts/prof_top.py $(ls $O/prof_solves_$v/*.db | head -1) 30 | grep -E "stream|winv|prep|kmat_gemv|trsv"
done
cd $R
echo "== determinism stress"; date
TGP_HIP_OPTIONS=stream_trsv=2 timeout 400 python scripts/stress_determinism.py 2>&1 | tail -6
date
} > $O/round.log 2>&1
tail -150 $O/round.log
