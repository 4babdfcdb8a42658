#!/bin/bash
# round 2, second session, batch 10: evidence of the final state (tests, bench lines, rocprofv3 kernel stats, PMC passes, timeline)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
B="--no-cpu-baseline --no-secondary"
{
echo "== pytest -m gpu"; date
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" $O/pytest_gpu.log | head -5
echo "== smoke"; date
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke
echo "== bench default"; date
timeout 600 python bench.py 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json; cut -c1-330 $O/bench_c2.json
echo "== torchrun launch line, one process"; date
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 5 --warmup 2 $B 2>/dev/null | tail -1 | cut -c1-300
echo "== other sizes"; date
for w in c1 n2048 n4096 n8192 n32768 n65536 ref2000 ref10000 ref20000; do timeout 400 python bench.py $B --workload $w --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_$w.json; cut -c1-250 $O/bench_$w.json; done
echo "== block-column path at world size 1 (c2, n65536)"; date
timeout 300 python bench.py --distributed --workload c2 --steps 10 --warmup 3 $B 2>/dev/null | tail -1 > $O/dist_c2.json; cut -c1-300 $O/dist_c2.json
timeout 300 python bench.py --distributed --workload n65536 --steps 2 --warmup 1 $B 2>/dev/null | tail -1 > $O/dist_n65536.json; cut -c1-300 $O/dist_n65536.json
echo "== adjacent paths"; date
timeout 200 python scripts/time_paths.py 16384 4096
timeout 200 python scripts/time_paths.py 4096 1024
echo "== rocprofv3 kernel stats"; date
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c2_final -o bench -- python $R/bench.py --steps 3 --warmup 1 $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_n65536_final -o bench -- python $R/bench.py --workload n65536 --steps 1 --warmup 1 $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solves -o bench -- python $R/scripts/time_paths.py 16384 4096 > /dev/null 2>&1
cd $R
for d in prof_c2_final prof_n65536_final; do echo "-- $d"; python scripts/prof_top.py $(ls $O/$d/*.db | head -1) 12; done
echo "-- prof_solves (scripts/time_paths.py 16384 4096)"; python scripts/prof_top.py $(ls $O/prof_solves/*.db | head -1) 40 | grep -E "stream|winv|prep|kmat|trsv|kgrad"
python scripts/timeline.py $(ls $O/prof_c2_final/*.db | head -1) /tmp/tl.csv 2500 > /dev/null; python scripts/timeline_panels.py /tmp/tl.csv | tail -16
echo "== PMC: fabric traffic (separate passes), c2 and the assembly at N = 65536"; date
cd /tmp
for cn in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc_$cn -o bench -- python $R/bench.py --steps 2 --warmup 1 $B --no-profile > /dev/null 2>&1
timeout 300 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc65_$cn -o bench -- python $R/bench.py --workload n65536 --steps 1 --warmup 0 $B --no-profile > /dev/null 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma -o bench -- python $R/bench.py --steps 2 --warmup 1 $B --no-profile > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc65_mfma -o bench -- python $R/bench.py --workload n65536 --steps 1 --warmup 0 $B --no-profile > /dev/null 2>&1
cd $R
for cn in FETCH_SIZE WRITE_SIZE; do echo "-- c2 $cn"; python scripts/pmc_summary.py $(ls $O/pmc_$cn/*.db | head -1) $cn | head -6; echo "-- n65536 $cn"; python scripts/pmc_summary.py $(ls $O/pmc65_$cn/*.db | head -1) $cn | head -8; done
echo "-- c2 MfmaUtil"; python scripts/pmc_multi.py $(ls $O/pmc_mfma/*.db | head -1) | head -8
echo "-- n65536 MfmaUtil"; python scripts/pmc_multi.py $(ls $O/pmc65_mfma/*.db | head -1) | head -8
echo "== chain variants (c2 / n8192 / n4096): default | no reserved slots | fused step | fused step + split gate"; date
for w in c2 n8192 n4096; do for o in "chain_reserve=128" "chain_reserve=0" "fused_step=1,gate_split=0" "fused_step=1,gate_split=1"; do echo "# $w $o"; TGP_HIP_OPTIONS="$o" timeout 300 python bench.py $B --workload $w --steps 8 --warmup 3 2>/dev/null | tail -1 | cut -c1-260; done; done
echo "== trailing-update kernel alone (K sweep)"; date
timeout 300 python scripts/gemm_bench.py f64 16384
echo "== determinism stress"; date
timeout 400 python scripts/stress_determinism.py 2>&1 | tail -6
date
} > $O/round.log 2>&1
tail -200 $O/round.log
