#!/bin/bash
# round 5, batch n: band order inside the block-cyclic trailing update, world size 1: TGP_HIP_OPTIONS reaches the driver's own context
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-r5n}
mkdir -p $O
export TMPDIR=/tmp
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
for b in 0 8 0 8; do TGP_HIP_OPTIONS=tile_band=$b timeout 300 python bench.py --distributed --workload c2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 tile_band=$b  %.2f ms' % d['ms_per_step'])"; done
for b in 0 8; do
TGP_HIP_OPTIONS=tile_band=$b timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_$b -o bench -- python bench.py --distributed --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo "-- tile_band=$b rc=$?"; python scripts/pmc_summary.py $(ls $O/pmc_$b/*.db | head -1) FETCH_SIZE | head -4 | tail -3
rm -rf $O/pmc_$b
done
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-200
