#!/bin/bash
# round 4: evidence batch on the current tree (one MI355X): full GPU suite, smoke, the driver's default bench line,
# other sizes, rocprofv3 kernel stats (c2 and n65536), PMC passes (fabric traffic of the trailing update, MFMA busy),
# the persistent chain's stamped timeline.  Output: gpurun_out/$1/log.txt (+ pmc_traffic.json, timelines).
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r4_final}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star"
{
echo "== pytest -m gpu"; date
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 2>&1 | tail -12
echo "== smoke"; date
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench default (the driver's line)"; date
timeout 600 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-600
echo "== other sizes"; date
for n in n1024 n2048 n4096 n8192 n32768; do timeout 300 python bench.py $B --workload $n --steps 10 2>/dev/null | tail -1 | tee -a $O/sizes.jsonl | cut -c1-330; done
echo "== the per-block chain (chain_kernel = 0) on the same box"; date
for n in n1024 n2048 n4096 n8192 c2; do timeout 300 python bench.py $B --workload $n --steps 10 --opt chain_kernel=0 2>/dev/null | tail -1 | tee -a $O/sizes_perblock.jsonl | cut -c1-330; done
echo "== rocprofv3 kernel stats, c2"; date
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o bench -- python bench.py --steps 10 --warmup 3 $B --no-profile > /dev/null 2>&1
python scripts/prof_top.py $(ls $O/kt_c2/*.db | head -1) 14
python scripts/timeline.py $(ls $O/kt_c2/*.db | head -1) $O/timeline_c2.csv 3000 > /dev/null; python scripts/timeline_dump.py $O/timeline_c2.csv > $O/timeline_c2.txt
rm -rf $O/kt_c2
echo "== rocprofv3 kernel stats, n65536 (2 steps + 1 warm-up)"; date
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_n65536 -o bench -- python bench.py --workload n65536 --steps 2 --warmup 1 $B --no-profile > /dev/null 2>&1
python scripts/prof_top.py $(ls $O/kt_n65536/*.db | head -1) 8
rm -rf $O/kt_n65536
echo "== PMC: fabric traffic of the trailing update (separate passes), c2"; date
for cn in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc_$cn -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile --opt chain_polls=0 > /dev/null 2>&1
echo "-- c2 $cn"; python scripts/pmc_summary.py $(ls $O/pmc_$cn/*.db | head -1) $cn | head -5
done
python scripts/pmc_to_bench.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) profiles/r04_e_final_evidence.md | cut -c1-400
cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
echo "== PMC: MFMA busy, c2"; date
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/pmc_mfma -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile --opt chain_polls=0 > /dev/null 2>&1
python scripts/pmc_multi.py $(ls $O/pmc_mfma/*.db | head -1) | head -6
rm -rf $O/pmc_mfma
echo "== persistent chain: stamped timeline (N = 1024, 4096)"; date
timeout 200 python scripts/chain_timeline.py 1024 4096 > $O/chain_timeline.txt 2>&1; head -14 $O/chain_timeline.txt
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-400 | tail -150
