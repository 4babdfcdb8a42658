#!/bin/bash
# round 5, batch r: the chain's ticket order, third form -- the whole diagonal lane of the next step (its solves, the two tiles
# the block after next builds on, that block's xsolve and potf2) in front of the bulk -- against batch q's form (DG only) and
# round 4's, three libraries on ONE box.
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-r5r}
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star --no-profile"
OLD=$R/tinygp_amd/lib/libtgp_hip_oldorder.so
V2=$R/tinygp_amd/lib/libtgp_hip_v2order.so
one() { timeout 300 python bench.py $B --workload $1 --steps $2 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.2f /s' % (d['ms_per_step'], d['value']))"; }
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== chain_check --quick (lane order)"
timeout 300 python scripts/chain_check.py --quick 2>&1 | tail -8
echo "== pytest subset (lane order)"; date
timeout 900 python -m pytest tests/test_gpu_0_kernels.py tests/test_gpu_1_gp.py -x -q -m gpu -p no:cacheprovider -k "variants or potrf or panel or stress or deterministic or config2 or config1 or mid_sizes or ragged or in_flight or indefinite or never_raises or timed_out" 2>&1 | tail -4
echo "== sizes: lane | DG-only (batch q) | round 4, same box, alternating"; date
for n in n2048 n4096 n8192 c2; do
  s=20; [ $n = c2 ] && s=12
  for rep in 1 2; do
  echo "-- $n lane"; one $n $s
  echo "-- $n dg";   TGP_HIP_LIBRARY=$V2 one $n $s
  echo "-- $n r4";   TGP_HIP_LIBRARY=$OLD one $n $s
  done
done
echo "== chain timeline, N = 4096 (lane order)"; date
timeout 200 python scripts/chain_timeline.py 4096 > $O/chain_timeline_lane.txt 2>&1; grep -A14 "^launch col" $O/chain_timeline_lane.txt | head -16
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-220
