#!/bin/bash
# round 5, batch c: RCCL from the C ABI (world size 1 through ncclBroadcast / ncclReduce / ncclAllReduce of the library,
# with and without torch in the process), peers on one GPU through the host-staged test transport, the bench contract;
# the timeout recovery path; the default options under rocprofv3 --pmc (tool detection); A/B of the polling host join.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5c}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star"
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== pytest: distributed path + new tests"
timeout 1200 python -m pytest tests/test_gpu_5_distributed.py tests/test_gpu_6_multirank_one_gpu.py tests/test_gpu_7_nccl_two_ranks.py tests/test_gpu_1_gp.py tests/test_gpu_0_kernels.py -x -q -m gpu -p no:cacheprovider --durations=6 -k "distributed or block or rccl or config4 or config5_dist or resident or seam or peers or two_rank or timed_out or poll_timeout or variants" 2>&1 | tail -16
echo "== pytest: bench contract"; date
timeout 900 python -m pytest tests/test_gpu_9_bench_contract.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
echo "== A/B: host join (polling with a deadline vs hipStreamSynchronize)"; date
for n in n1024 n4096 c2; do for hj in 0 1 0 1; do
echo "-- $n host_join=$hj"
timeout 300 python bench.py $B --no-profile --workload $n --steps 20 --opt host_join=$hj 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
echo "== default options under rocprofv3 --pmc (ROCPROF_COUNTER_COLLECTION -> chain_polls = 0)"; date
for cn in FETCH_SIZE WRITE_SIZE; do
timeout 150 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc_$cn -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile > $O/pmc_$cn.log 2>&1
echo "-- c2 $cn rc=$?"; tail -1 $O/pmc_$cn.log | cut -c1-160; python scripts/pmc_summary.py $(ls $O/pmc_$cn/*.db | head -1) $cn | head -5
done
python scripts/pmc_to_bench.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) profiles/r05_c | cut -c1-400
cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/pmc_mfma -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile > $O/pmc_mfma.log 2>&1
echo "-- MFMA busy rc=$?"; python scripts/pmc_multi.py $(ls $O/pmc_mfma/*.db | head -1) | head -8
rm -rf $O/pmc_mfma
echo "== bench default"; date
timeout 600 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-700
date
} > $O/log.txt 2>&1
tail -120 $O/log.txt | cut -c1-400
