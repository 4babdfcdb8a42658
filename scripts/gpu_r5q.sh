#!/bin/bash
# round 5, batch q: the chain's ticket order -- the next step's diagonal tasks in front of the bulk updates -- against round 4's
# order (a second library built from the old csrc/chain_tasks.h, TGP_HIP_LIBRARY) on ONE box: correctness, sizes, timeline.
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-r5q}
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star --no-profile"
OLD=$R/tinygp_amd/lib/libtgp_hip_oldorder.so
one() { timeout 300 python bench.py $B --workload $1 --steps $2 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.2f /s' % (d['ms_per_step'], d['value']))"; }
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== chain_check --quick (new order)"
timeout 300 python scripts/chain_check.py --quick 2>&1 | tail -8
echo "== pytest subset (new order)"; date
timeout 900 python -m pytest tests/test_gpu_0_kernels.py tests/test_gpu_1_gp.py -x -q -m gpu -p no:cacheprovider -k "variants or potrf or panel or stress or deterministic or config2 or config1 or mid_sizes or ragged or in_flight or indefinite or never_raises or timed_out" 2>&1 | tail -4
echo "== sizes: new order | old order (same box, alternating)"; date
for n in n1024 n2048 n4096 n8192 c2; do
  s=20; [ $n = c2 ] && s=12
  echo "-- $n new"; one $n $s
  echo "-- $n old"; TGP_HIP_LIBRARY=$OLD one $n $s
  echo "-- $n new"; one $n $s
  echo "-- $n old"; TGP_HIP_LIBRARY=$OLD one $n $s
done
echo "-- n32768 new"; one n32768 4; echo "-- n32768 old"; TGP_HIP_LIBRARY=$OLD one n32768 4
echo "== chain timeline, N = 4096 (new order)"; date
timeout 200 python scripts/chain_timeline.py 4096 > $O/chain_timeline_new.txt 2>&1; grep -A14 "^launch col" $O/chain_timeline_new.txt | head -16
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-220
