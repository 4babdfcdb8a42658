#!/bin/bash
# round 5, batch k: fabric traffic of the trailing update by band height (tile_band), c2
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-r5k}
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star"
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
for b in 4 8 12 24 32 64; do
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_$b -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile --opt tile_band=$b > /dev/null 2>&1
echo "-- tile_band=$b FETCH_SIZE rc=$?"; python scripts/pmc_summary.py $(ls $O/pmc_$b/*.db | head -1) FETCH_SIZE | head -3 | tail -2
rm -rf $O/pmc_$b
done
echo "== times"; date
for b in 0 8 16 8 0 16; do timeout 300 python bench.py $B --steps 20 --no-profile --opt tile_band=$b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tile_band=$b  %.3f ms' % d['ms_per_step'])"; done
for b in 0 8 16; do timeout 300 python bench.py $B --workload n65536 --steps 2 --warmup 1 --opt tile_band=$b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n65536 tile_band=$b  %.1f ms  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; done
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-200
