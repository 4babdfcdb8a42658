#!/bin/bash
# round 5, batch h: A/B of the deferred side assembly (asm_defer) and of big-tile gates with the split tail; the block-column
# bench line at world size 1 over nccl (RcclComm.from_torch); parity subset with asm_defer on.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5h}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star --no-profile"
one() { timeout 300 python bench.py $B --workload $1 --steps 20 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.2f /s' % (d['ms_per_step'], d['value']))"; }
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
for rep in 1 2; do
echo "-- c2 default";                              one c2
echo "-- c2 asm_defer=1";                          one c2 --opt asm_defer=1
echo "-- c2 first_small_tiles=300 split_tail=1";   one c2 --opt first_small_tiles=300 --opt split_tail=1
echo "-- c2 asm_defer=1 nb_first=512";             one c2 --opt asm_defer=1 --opt nb_first=512
done
for rep in 1 2; do
echo "-- n8192 default";     one n8192
echo "-- n8192 asm_defer=1"; one n8192 --opt asm_defer=1
done
echo "== parity subset with asm_defer=1"; date
TGP_HIP_OPTIONS=asm_defer=1 timeout 600 python -m pytest tests/test_gpu_1_gp.py -x -q -m gpu -p no:cacheprovider -k "config2 or mid_sizes or ragged or deterministic or refactor or config3" 2>&1 | tail -4
echo "== block-column bench line, world size 1 over nccl (the id through torch, the collectives by the library)"; date
timeout 600 python bench.py --distributed --workload n8192 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
date
} > $O/log.txt 2>&1
tail -60 $O/log.txt | cut -c1-400
