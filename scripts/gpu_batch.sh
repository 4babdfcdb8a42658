#!/bin/bash
# ONE parameterised GPU batch (replaces round 5's gpu_r5[a-r].sh): `gpu_batch.sh <tag> <section> [<section> ...]`.
# Every section appends to gpurun_out/<tag>/log.txt; the tail of the log is what the gpurun call prints.
#   suite        pytest -m gpu in the driver's order + smoke
#   sizes        N = 1 024 ... 8 192 and c2, ms per evaluation (no profiler), with $OPTS (e.g. "chain_full_rows=8192")
#   ab:<opts>    the same sizes list with the given comma-separated options, interleaved with the default (A/B on one box)
#   stats        rocprofv3 --kernel-trace --stats of the bench command (c2)
#   pmc          FETCH_SIZE / WRITE_SIZE passes + the traffic stamp
#   whole        whole-evaluation counters (SQ_BUSY_CU_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE ...), c2
#   spans        span dump of the trailing-update launches (no profiler)
#   timeline     kernel timeline of one c2 evaluation (first / last 2.5 ms)
#   chain:<n>    stamped timeline of the chain launch at N = n
#   bench        the driver's default line
#   py:<script and args, '+' for spaces>   any helper under scripts/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
TAG=${1:-r6}
shift
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star"
SIZES=${SIZES:-"n1024 n2048 n4096 n8192 c2"}
one() {  # workload steps opts -> "x.xxx ms"
  local o=""
  for kv in $(echo "$3" | tr ',' ' '); do o="$o --opt $kv"; done
  timeout 300 python bench.py $B --no-profile --workload $1 --steps $2 --warmup 3 $o 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['ms_per_step'])"
}
steps_of() { case $1 in c2) echo 12;; n8192) echo 30;; *) echo 60;; esac; }
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)  tag $TAG"; date
for sec in "$@"; do
case $sec in
suite)
  echo "== pytest tests -x -q -m gpu (driver order)"
  timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=6 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -16
  echo "== smoke"; date
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 ;;
sizes)
  echo "== sizes, options: ${OPTS:-default}"
  for n in $SIZES; do echo "-- $n $(one $n $(steps_of $n) "$OPTS")"; done ;;
ab:*)
  V=${sec#ab:}
  echo "== A/B on this box: default | $V"
  for n in $SIZES; do for rep in 1 2; do
    echo "-- $n default $(one $n $(steps_of $n) "")   |   $V $(one $n $(steps_of $n) "$V")"
  done; done ;;
stats)
  echo "== rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 (c2) ${OPTS}"
  o=""; for kv in $(echo "$OPTS" | tr ',' ' '); do o="$o --opt $kv"; done
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o bench -- python bench.py --steps 10 --warmup 3 $B --no-profile $o > /dev/null 2>&1
  python scripts/prof_top.py $(ls $O/kt_c2/*.db | head -1) 16
  rm -rf $O/kt_c2 ;;
pmc)
  echo "== PMC passes, default options"
  for cn in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc_$cn -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile > $O/pmc_$cn.log 2>&1
    echo "-- c2 $cn rc=$?"; python scripts/pmc_summary.py $(ls $O/pmc_$cn/*.db | head -1) $cn | head -5
  done
  python scripts/pmc_to_bench.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) profiles/${TAG}_final_evidence.md | cut -c1-400
  cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
whole)
  echo "== whole-evaluation counters, c2 (every kernel of 2 evaluations + 1 warm-up; one counter set per pass)"
  for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES"; do
    tagc=$(echo $set | tr ' ' '_')
    timeout 240 rocprofv3 --pmc $set --kernel-trace -d $O/w_$tagc -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile > $O/w_$tagc.log 2>&1
    echo "-- $set rc=$?"
    python scripts/pmc_whole.py $(ls $O/w_$tagc/*.db | head -1) $set
    rm -rf $O/w_$tagc
  done ;;
spans)
  echo "== span dump (trailing-update launches, no profiler) ${OPTS}"
  o=""; for kv in $(echo "$OPTS" | tr ',' ' '); do o="$o --opt $kv"; done
  TGP_SPAN_DUMP=1 timeout 300 python bench.py $B --steps 2 --warmup 2 $o 2>&1 >/dev/null | grep "^span" | tail -12 ;;
timeline)
  echo "== kernel timeline of one c2 evaluation ${OPTS}"
  o=""; for kv in $(echo "$OPTS" | tr ',' ' '); do o="$o --opt $kv"; done
  timeout 300 rocprofv3 --kernel-trace -d $O/tl -o bench -- python bench.py --steps 3 --warmup 2 $B --no-profile $o > /dev/null 2>&1
  python scripts/timeline.py $(ls $O/tl/*.db | head -1) $O/timeline.csv 2500 > /dev/null 2>&1
  python scripts/timeline_dump.py $O/timeline.csv 2>&1 | head -${TL_LINES:-150}
  rm -rf $O/tl ;;
chain:*)
  n=${sec#chain:}
  echo "== stamped chain timeline, N = $n ${OPTS}"
  timeout 300 python scripts/chain_timeline.py $n $OPTS 2>&1 | tail -80 ;;
bench)
  echo "== bench default (the driver's line)"
  timeout 1200 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-1800 ;;
py:*)
  cmd=$(echo "${sec#py:}" | tr '+' ' ')
  echo "== python scripts/$cmd"
  timeout 900 python scripts/$cmd 2>&1 | tail -60 ;;
*) echo "unknown section $sec" ;;
esac
date
done
} > $O/log.txt 2>&1
tail -150 $O/log.txt | cut -c1-420
