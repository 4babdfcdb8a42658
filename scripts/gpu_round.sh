#!/bin/bash
# round 2, batch 1: full GPU suite, default bench, block-column path at world size 1,
# rocprofv3 kernel stats at N = 16384 / 65536 and the MFMA-utilisation counters at N = 65536
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
DESEL=""
python -c "import numpy as np,sys; sys.exit(0 if 'c3_n65536__logp' in np.load('tests/golden/large.npz').files else 1)" 2>/dev/null || DESEL="--deselect tests/test_gpu_gp.py::test_config3_n65536_full_size"
python -c "import numpy as np,sys; sys.exit(0 if 'c5_n32768__logp' in np.load('tests/golden/large.npz').files else 1)" 2>/dev/null || DESEL="$DESEL --deselect tests/test_gpu_gp.py::test_config5_kernel_fp32_posterior_mean_n32768 --deselect tests/test_gpu_distributed.py::test_config5_distributed_condition_mean_fp32"
{
echo "== pytest -m gpu"; date
timeout 1200 python -m pytest tests -m gpu -q $DESEL 2>&1 | tail -25
echo "== bench default"; date
timeout 400 python bench.py 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json; cut -c1-700 $O/bench_c2.json
echo "== block-column path, world size 1"; date
timeout 300 python bench.py --distributed --workload c2 --steps 10 --warmup 3 2>$O/dist_c2.err | tail -1 > $O/dist_c2.json; cut -c1-400 $O/dist_c2.json
timeout 300 python bench.py --distributed --workload n65536 --steps 2 --warmup 1 2>$O/dist_n65536.err | tail -1 > $O/dist_n65536.json; cut -c1-400 $O/dist_n65536.json
timeout 300 python bench.py --workload n65536 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/single_n65536.json; cut -c1-400 $O/single_n65536.json
echo "== reference recipe"; date
for n in 2000 10000 20000; do timeout 200 python bench.py --workload ref$n --steps 10 --warmup 3 --no-secondary 2>/dev/null | tail -1 > $O/bench_ref$n.json; cut -c1-300 $O/bench_ref$n.json; done
echo "== N sweep"; date
for n in 4096 8192 32768; do timeout 200 python bench.py --workload n$n --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-260; done
echo "== rocprofv3 kernel stats"; date
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_n65536 -o bench -- python $R/bench.py --workload n65536 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_n65536.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dist_c2 -o bench -- python $R/bench.py --distributed --workload c2 --steps 3 --warmup 1 > $O/prof_dist_c2.log 2>&1
cd $R
for d in prof_c2 prof_n65536 prof_dist_c2; do echo "-- $d"; python scripts/prof_top.py $(ls $O/$d/*.db | head -1) 14; done
echo "== PMC: MFMA utilisation at N = 65536"; date
rocprofv3 -L 2>/dev/null | grep -E "SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE|SQ_INSTS_VALU_MFMA_MOPS_F64|SQ_BUSY_CU_CYCLES" | head -10
cd /tmp
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma_n65536 -o bench -- python $R/bench.py --workload n65536 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-profile > $O/pmc_mfma_n65536.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma_c2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-profile > $O/pmc_mfma_c2.log 2>&1
cd $R
for d in pmc_mfma_n65536 pmc_mfma_c2; do echo "-- $d"; python scripts/pmc_multi.py $(ls $O/$d/*.db | head -1) | head -12; done
date
} > $O/round.log 2>&1
tail -150 $O/round.log
