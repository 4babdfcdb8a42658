#!/bin/bash
# round 2, batch 3: backward streaming solve, big-tile in-panel updates (sweep), NB = 2048 at large N,
# VALU instruction counts of the assembly kernel, adjacent-path timings, full-N cpu baseline
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
DESEL=""
python -c "import numpy as np,sys; sys.exit(0 if 'c3_n65536__logp' in np.load('tests/golden/large.npz').files else 1)" 2>/dev/null || DESEL="--deselect tests/test_gpu_gp.py::test_config3_n65536_full_size"
python -c "import numpy as np,sys; sys.exit(0 if 'c5_n32768__logp' in np.load('tests/golden/large.npz').files else 1)" 2>/dev/null || DESEL="$DESEL --deselect tests/test_gpu_gp.py::test_config5_kernel_fp32_posterior_mean_n32768 --deselect tests/test_gpu_distributed.py::test_config5_distributed_condition_mean_fp32"
B="--no-cpu-baseline --no-secondary"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('$1', 'evals/s %.2f ms %.2f  update %.1f TF (%.3f)  potrf %.2f ms' % (d['value'], d['ms_per_step'], r.get('achieved',0), r.get('frac',0), (d.get('stage_ms') or {}).get('potrf',0)))"; }
{
echo "== pytest -m gpu (defaults)"; date
timeout 1200 python -m pytest tests -m gpu -q -x $DESEL 2>&1 | tail -12
echo "== in-panel updates on 128x128 tiles, config 2"; date
for t in 0 32 64 128 256 512; do
  TGP_HIP_OPTIONS="inpanel_big_min_tiles=$t" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [inpanel_big_min_tiles=$t]"
done
for w in n4096 n8192 n32768; do for t in 0 64 256; do
  TGP_HIP_OPTIONS="inpanel_big_min_tiles=$t" timeout 120 python bench.py $B --workload $w --steps 5 --warmup 2 2>/dev/null | tail -1 | line "$w [inpanel_big_min_tiles=$t]"
done; done
echo "== outer block 2048 at large N"; date
for w in n32768 n65536; do for opts in "nb_outer=1024" "nb_outer=2048" "nb_outer=2048,inpanel_big_min_tiles=128"; do
  TGP_HIP_OPTIONS="$opts" timeout 200 python bench.py $B --workload $w --steps 2 --warmup 1 2>/dev/null | tail -1 | line "$w [$opts]"
done; done
TGP_HIP_OPTIONS="nb_outer=2048" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [nb_outer=2048]"
TGP_HIP_OPTIONS="nb_outer=512" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [nb_outer=512]"
echo "== adjacent paths (scripts/time_paths.py)"; date
timeout 200 python scripts/time_paths.py 16384 4096
timeout 100 python scripts/time_paths.py 4096 1024
echo "== default bench: secondary rooflines + cpu baseline at the workload's N"; date
timeout 600 python bench.py 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json; cut -c1-300 $O/bench_c2.json
python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(json.dumps(d.get('roofline_secondary'))[:700]); print(json.dumps(d.get('cpu_baseline'))[:1100])"
echo "== VALU instruction mix of the assembly kernel (PMC)"; date
cd /tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_kmat -o bench -- python $R/bench.py --workload n16384 --steps 1 --warmup 0 $B --no-profile > $O/pmc_kmat.log 2>&1
cd $R
python scripts/pmc_multi.py $(ls $O/pmc_kmat/*.db | head -1) | grep -E "^#|kmat"
date
} > $O/round.log 2>&1
tail -100 $O/round.log
