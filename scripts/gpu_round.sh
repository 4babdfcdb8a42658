#!/bin/bash
# round 2, batch 6: eight-wave fp64 trailing-update kernel (gemm8) vs the four-wave one
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
DESEL=""
python -c "import numpy as np,sys; sys.exit(0 if 'c3_n65536__logp' in np.load('tests/golden/large.npz').files else 1)" 2>/dev/null || DESEL="--deselect tests/test_gpu_gp.py::test_config3_n65536_full_size"
B="--no-cpu-baseline --no-secondary"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('$1', 'evals/s %.2f ms %.2f  update %.1f TF (%.3f)  potrf %.2f ms' % (d['value'], d['ms_per_step'], r.get('achieved',0), r.get('frac',0), (d.get('stage_ms') or {}).get('potrf',0)))"; }
{
echo "== parity with gemm8=1"; date
TGP_HIP_OPTIONS="gemm8=1" timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py tests/test_gpu_distributed.py -m gpu -q -x $DESEL -k "potrf or ragged or config or determin or mid_sizes or golden or block_cyclic or ill_cond" 2>&1 | tail -5
echo "== gemm8 A/B"; date
for rep in 1 2; do for g8 in 0 1; do
  TGP_HIP_OPTIONS="gemm8=$g8" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [gemm8=$g8]"
done; done
for w in n8192 n32768; do for g8 in 0 1; do
  TGP_HIP_OPTIONS="gemm8=$g8" timeout 120 python bench.py $B --workload $w --steps 5 --warmup 2 2>/dev/null | tail -1 | line "$w [gemm8=$g8]"
done; done
for g8 in 0 1; do
  TGP_HIP_OPTIONS="gemm8=$g8" timeout 200 python bench.py $B --workload n65536 --steps 2 --warmup 1 2>/dev/null | tail -1 | line "n65536 [gemm8=$g8]"
done
TGP_HIP_OPTIONS="gemm8=1,nb_outer=2048" timeout 200 python bench.py $B --workload n65536 --steps 2 --warmup 1 2>/dev/null | tail -1 | line "n65536 [gemm8=1,nb_outer=2048]"
TGP_HIP_OPTIONS="gemm8=1,lookahead=0" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [gemm8=1,lookahead=0]"
TGP_HIP_OPTIONS="gemm8=0,lookahead=0" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [gemm8=0,lookahead=0]"
echo "== kernel stats gemm8=1"; date
cd /tmp
TGP_HIP_OPTIONS="gemm8=1" timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_c2_g8 -o bench -- python $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2>&1
TGP_HIP_OPTIONS="gemm8=1" timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_n65536_g8 -o bench -- python $R/bench.py --workload n65536 --steps 1 --warmup 0 $B > /dev/null 2>&1
cd $R
for d in prof_c2_g8 prof_n65536_g8; do echo "-- $d"; python scripts/prof_top.py $(ls $O/$d/*.db | head -1) 6; done
date
} > $O/round.log 2>&1
tail -60 $O/round.log
