#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
for s in 0 2 3 4 6; do echo "== prefetch $s"; TGP_HIP_OPTIONS=gemm_prefetch=$s timeout 120 python scripts/gemm_bench.py f64 16384 | grep mode; done
for s in 0 2 4; do echo "== c2 prefetch $s"; TGP_HIP_OPTIONS=gemm_prefetch=$s bash scripts/bench_variants.sh "" | tail -1; done
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
