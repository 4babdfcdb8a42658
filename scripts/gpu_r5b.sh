#!/bin/bash
# round 5, batch b: update tasks on the 4x4x4 MFMA form (chain_fast_update) and the stream wait-value followers
# (chain_polls = 1) -- correctness first, then A/B benches on ONE box, the default path under rocprofv3 --pmc with no
# --opt, stamped chain timelines.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5b}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star"
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== chain_check --quick"
timeout 300 python scripts/chain_check.py --quick 2>&1 | tail -25
echo "== pytest subset"; date
timeout 900 python -m pytest tests/test_gpu_0_kernels.py tests/test_gpu_1_gp.py -x -q -m gpu -p no:cacheprovider -k "variants or potrf or panel or split_tail or stress or deterministic or config2 or config1 or mid_sizes or ragged or in_flight or indefinite or never_raises" 2>&1 | tail -6
echo "== A/B on this box: chain_fast_update x chain_polls"; date
for n in n4096 c2; do for fu in 0 1; do for po in 2 1; do
echo "-- $n fast_update=$fu polls=$po"
timeout 300 python bench.py $B --workload $n --steps 12 --opt chain_fast_update=$fu --opt chain_polls=$po 2>/dev/null | tail -1 | tee -a $O/ab.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))"
done; done; done
for n in n1024 n2048 n8192; do for fu in 0 1; do
echo "-- $n fast_update=$fu"
timeout 300 python bench.py $B --no-profile --workload $n --steps 12 --opt chain_fast_update=$fu 2>/dev/null | tail -1 | tee -a $O/ab.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
for fu in 0 1; do
echo "-- n65536 fast_update=$fu"
timeout 300 python bench.py $B --workload n65536 --steps 2 --warmup 1 --opt chain_fast_update=$fu 2>/dev/null | tail -1 | tee -a $O/ab.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))"
done
echo "== default path under rocprofv3 --pmc, NO --opt (item 5)"; date
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/pmc_mfma -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile > $O/pmc_run.log 2>&1
echo "rc=$?"; tail -1 $O/pmc_run.log | cut -c1-200
python scripts/pmc_multi.py $(ls $O/pmc_mfma/*.db | head -1) | head -8
rm -rf $O/pmc_mfma
echo "== chain timelines (N = 1024, 4096), fast update on / off"; date
timeout 200 python scripts/chain_timeline.py 1024 4096 > $O/chain_timeline_fast.txt 2>&1; grep -A12 "update" $O/chain_timeline_fast.txt | tail -30
TGP_HIP_OPTIONS=chain_fast_update=0 timeout 200 python scripts/chain_timeline.py 4096 > $O/chain_timeline_old.txt 2>&1; grep -A12 "update" $O/chain_timeline_old.txt | tail -16
date
} > $O/log.txt 2>&1
tail -150 $O/log.txt | cut -c1-400
