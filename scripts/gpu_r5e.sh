#!/bin/bash
# round 5, batch e: kernel timeline + per-kernel stats of the default schedule (c2 and n65536), other sizes, secondary rooflines.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5e}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star"
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== rocprofv3 kernel stats, c2"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o bench -- python bench.py --steps 10 --warmup 3 $B --no-profile > /dev/null 2>&1
python scripts/prof_top.py $(ls $O/kt_c2/*.db | head -1) 14
python scripts/timeline.py $(ls $O/kt_c2/*.db | head -1) $O/timeline_c2.csv 3000 > /dev/null; python scripts/timeline_dump.py $O/timeline_c2.csv > $O/timeline_c2.txt
rm -rf $O/kt_c2
echo "== span dump (where each trailing-update launch sits, no profiler)"; date
TGP_SPAN_DUMP=1 timeout 200 python bench.py $B --steps 3 --warmup 2 2>&1 | grep "span at" | tail -12
echo "== rocprofv3 kernel stats, n65536 (2 steps + 1 warm-up)"; date
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_n65536 -o bench -- python bench.py --workload n65536 --steps 2 --warmup 1 $B --no-profile > /dev/null 2>&1
python scripts/prof_top.py $(ls $O/kt_n65536/*.db | head -1) 8
rm -rf $O/kt_n65536
echo "== sizes"; date
for n in n1024 n2048 n4096 n8192 n32768; do timeout 300 python bench.py $B --no-profile --workload $n --steps 10 2>/dev/null | tail -1 | tee -a $O/sizes.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['n'], d['ms_per_step'], d['value'])"; done
echo "== bench default with secondary rooflines (no north star)"; date
timeout 600 python bench.py --no-north-star 2>/dev/null | tail -1 | tee $O/bench_secondary.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], json.dumps(d.get('roofline')), json.dumps(d.get('roofline_secondary'))[:1500])"
date
} > $O/log.txt 2>&1
tail -100 $O/log.txt | cut -c1-400
