#!/bin/bash
# round 5, batch m: band order also inside the block-cyclic trailing update (launch_gemm_nt_dist): parity of the block-column
# path, time and fabric traffic at world size 1.
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-r5m}
mkdir -p $O
export TMPDIR=/tmp
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
timeout 900 python -m pytest tests/test_gpu_5_distributed.py tests/test_gpu_6_multirank_one_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4
echo "== block-column log_probability, world size 1 (bench --distributed), tile_band 0 / 8"; date
for b in 0 8 0 8; do timeout 300 python bench.py --distributed --workload c2 --steps 8 --warmup 2 --no-cpu-baseline --opt tile_band=$b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 tile_band=$b  %.2f ms' % d['ms_per_step'])"; done
for b in 0 8; do timeout 300 python bench.py --distributed --workload n65536 --steps 2 --warmup 1 --no-cpu-baseline --opt tile_band=$b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n65536 tile_band=$b  %.1f ms' % d['ms_per_step'])"; done
echo "== FETCH_SIZE of the block-cyclic update, c2, tile_band 0 / 8"; date
for b in 0 8; do
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_$b -o bench -- python bench.py --distributed --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --opt tile_band=$b > /dev/null 2>&1
echo "-- tile_band=$b rc=$?"; python scripts/pmc_summary.py $(ls $O/pmc_$b/*.db | head -1) FETCH_SIZE | head -4 | tail -3
rm -rf $O/pmc_$b
done
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-200
