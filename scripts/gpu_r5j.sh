#!/bin/bash
# round 5, batch j: band order of the MFMA products' tiles (ctx option tile_band): time and fabric traffic against the column order.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5j}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-north-star"
one() { timeout 300 python bench.py $B --workload $1 --steps $2 --warmup $3 "${@:4}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('%.3f ms  %.2f /s  frac %s  launch %s ms' % (d['ms_per_step'], d['value'], r.get('frac'), r.get('avg_launch_ms')))"; }
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== parity subset with tile_band=16"
TGP_HIP_OPTIONS=tile_band=16 timeout 600 python -m pytest tests/test_gpu_0_kernels.py tests/test_gpu_1_gp.py -x -q -m gpu -p no:cacheprovider -k "gemm or potrf or variants or config2 or mid_sizes or ragged or deterministic or config3 or split_tail" 2>&1 | tail -4
echo "== c2"; date
for rep in 1 2; do for b in 0 8 16 32; do echo "-- c2 tile_band=$b"; one c2 12 3 --opt tile_band=$b; done; done
echo "== n8192 / n32768"; date
for b in 0 16; do echo "-- n8192 tile_band=$b"; one n8192 12 3 --opt tile_band=$b --no-profile; done
for b in 0 16 32; do echo "-- n32768 tile_band=$b"; one n32768 4 1 --opt tile_band=$b; done
echo "== n65536"; date
for b in 0 16 32; do echo "-- n65536 tile_band=$b"; one n65536 2 1 --opt tile_band=$b; done
echo "== fabric traffic (PMC FETCH_SIZE, WRITE_SIZE), c2, tile_band 0 / 16"; date
for b in 0 16; do for cn in FETCH_SIZE WRITE_SIZE; do
timeout 150 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc_${b}_$cn -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile --opt tile_band=$b > /dev/null 2>&1
echo "-- tile_band=$b $cn rc=$?"; python scripts/pmc_summary.py $(ls $O/pmc_${b}_$cn/*.db | head -1) $cn | head -4
rm -rf $O/pmc_${b}_$cn
done; done
date
} > $O/log.txt 2>&1
tail -80 $O/log.txt | cut -c1-300
