#!/bin/bash
# round 3, batch 18: assembly kernel shapes by size; does the run-time patch flag cost the trailing-update kernels anything (A/B libraries)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b18
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline"
sec() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:12], round(d['value'],4), round(d['ms_per_step'],3))
for r in d.get('roofline_secondary', [])[:1]: print('   ', r['kernel'][:40], round(r['achieved'],1), r['unit'], 'ms', r.get('ms'))"; }
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['workload'][:14], d['value'], d['ms_per_step'], 'update TF', round(r['achieved'],2), 'frac', round(r['frac'],4))"; }
{
date
for sh in 0 1 2; do
echo "== kmat_shape=$sh"
TGP_HIP_OPTIONS=kmat_shape=$sh timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "kmat or kernel_matrix or assembly" 2>&1 | tail -1
TGP_HIP_OPTIONS=kmat_shape=$sh timeout 300 python bench.py --workload n4096 --steps 3 --warmup 1 $B 2>/dev/null | tail -1 | sec
TGP_HIP_OPTIONS=kmat_shape=$sh timeout 300 python bench.py --steps 3 --warmup 1 $B 2>/dev/null | tail -1 | sec
TGP_HIP_OPTIONS=kmat_shape=$sh timeout 300 python bench.py --workload n32768 --steps 2 --warmup 1 $B 2>/dev/null | tail -1 | sec
TGP_HIP_OPTIONS=kmat_shape=$sh timeout 300 python bench.py --workload n65536 --steps 1 --warmup 1 $B 2>/dev/null | tail -1 | sec
done
echo "== A/B: run-time patch flag compiled in (A) / out (B)"
for rep in 1 2; do
for lib in tinygp_amd/lib build_ab; do
echo "-- $lib"
TGP_HIP_LIBRARY=$R/$lib/libtgp_hip.so timeout 300 python bench.py --workload n65536f32 --steps 3 --warmup 1 $B --no-secondary 2>/dev/null | tail -1 | line
TGP_HIP_LIBRARY=$R/$lib/libtgp_hip.so timeout 300 python bench.py --workload n131072f32 --steps 2 --warmup 1 $B --no-secondary 2>/dev/null | tail -1 | line
TGP_HIP_LIBRARY=$R/$lib/libtgp_hip.so timeout 300 python bench.py --steps 10 --warmup 3 $B --no-secondary 2>/dev/null | tail -1 | line
done
done
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-300
