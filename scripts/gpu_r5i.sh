#!/bin/bash
# round 5, batch i (final tree): the WHOLE GPU suite in driver order, smoke, rocprofv3 kernel stats of the bench command,
# PMC passes with the default options (traffic stamp), the default bench line.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5i}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star"
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== pytest tests -x -q -m gpu (driver order)"
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=6 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -16
echo "== smoke"; date
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 (c2)"; date
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o bench -- python bench.py --steps 10 --warmup 3 $B --no-profile > /dev/null 2>&1
python scripts/prof_top.py $(ls $O/kt_c2/*.db | head -1) 14
rm -rf $O/kt_c2
echo "== PMC passes, default options"; date
for cn in FETCH_SIZE WRITE_SIZE; do
timeout 150 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc_$cn -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile > $O/pmc_$cn.log 2>&1
echo "-- c2 $cn rc=$?"; python scripts/pmc_summary.py $(ls $O/pmc_$cn/*.db | head -1) $cn | head -4
done
python scripts/pmc_to_bench.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) profiles/r05_s_final_evidence.md | cut -c1-300
cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
echo "== ticket orders on this box: hybrid (tree) | lane everywhere (batch r's library is gone: skipped) | DG-only | round 4"; date
one() { timeout 300 python bench.py $B --no-profile --workload $1 --steps $2 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['ms_per_step'])"; }
for n in n4096 c2; do s=20; [ $n = c2 ] && s=12; for rep in 1 2; do
echo "-- $n hybrid"; one $n $s
[ -f tinygp_amd/lib/libtgp_hip_v2order.so ] && { echo "-- $n dg-only"; TGP_HIP_LIBRARY=$R/tinygp_amd/lib/libtgp_hip_v2order.so one $n $s; }
[ -f tinygp_amd/lib/libtgp_hip_oldorder.so ] && { echo "-- $n round4"; TGP_HIP_LIBRARY=$R/tinygp_amd/lib/libtgp_hip_oldorder.so one $n $s; }
done; done
echo "== bench default (the driver's line)"; date
timeout 900 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-1500
date
} > $O/log.txt 2>&1
tail -70 $O/log.txt | cut -c1-400
