#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -3
for o in 0 1 0 1; do for w in c2 n8192; do echo "## supertiles=$o $w"; TGP_HIP_OPTIONS=gemm_supertiles=$o bash scripts/bench_variants.sh "--workload $w" | tail -1 | cut -c1-70; done; done
for o in 0 1; do echo "## supertiles=$o n32768"; TGP_HIP_OPTIONS=gemm_supertiles=$o bash scripts/bench_variants.sh "--workload n32768 --steps 3 --warmup 1" | tail -1 | cut -c1-70; done
cd /tmp
for C in FETCH_SIZE; do
rm -rf /tmp/p1; timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/p1 -o pm -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | grep '"metric"' | cut -c1-100
python $R/scripts/pmc_summary.py $(find /tmp/p1 -name "*.db" | head -1) $C | head -4 | tee $R/gpurun_out/pmc_$C.txt
done
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
