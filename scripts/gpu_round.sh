#!/bin/bash
# round 2, batch 2: new solve kernels + chain-protection options (sweep) + block-column path again
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
echo "== option sweep, config 2"; date
for opts in "" "trsm_split=1" "reserve_cus=8" "reserve_cus=8,trsm_split=1" "reserve_cus=16,trsm_split=1" "reserve_cus=4,trsm_split=1" "epi_atomic=1" "reserve_cus=8,trsm_split=1,epi_atomic=1" "reserve_cus=8,trsm_split=1,first_split=0"; do
  TGP_HIP_OPTIONS="$opts" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [$opts]"
done
echo "== option sweep, other sizes"; date
for w in n4096 n8192 n32768; do for opts in "" "reserve_cus=8,trsm_split=1" "reserve_cus=8,trsm_split=1,epi_atomic=1"; do
  TGP_HIP_OPTIONS="$opts" timeout 120 python bench.py $B --workload $w --steps 5 --warmup 2 2>/dev/null | tail -1 | line "$w [$opts]"
done; done
for opts in "" "epi_atomic=1" "reserve_cus=8,trsm_split=1,epi_atomic=1"; do
  TGP_HIP_OPTIONS="$opts" timeout 120 python bench.py $B --workload n65536 --steps 1 --warmup 1 2>/dev/null | tail -1 | line "n65536 [$opts]"
done
echo "== parity under the options"; date
TGP_HIP_OPTIONS="reserve_cus=8,trsm_split=1,epi_atomic=1" timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py tests/test_gpu_grad.py -m gpu -q -x $DESEL -k "potrf or ragged or config2 or config1 or determin or mid_sizes or grad or golden" 2>&1 | tail -5
echo "== block-column path, world size 1"; date
timeout 300 python bench.py --distributed --workload c2 --steps 10 --warmup 3 2>$O/dist_c2.err | tail -1 > $O/dist_c2.json; cat $O/dist_c2.json | line "dist c2"
TGP_HIP_OPTIONS="reserve_cus=8,trsm_split=1" timeout 300 python bench.py --distributed --workload c2 --steps 10 --warmup 3 2>/dev/null | tail -1 | line "dist c2 [reserve+split]"
echo "== default bench with secondary rooflines + cpu baseline"; date
timeout 400 python bench.py 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json; cut -c1-300 $O/bench_c2.json
python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(json.dumps(d.get('roofline_secondary'))[:900]); print(json.dumps(d.get('cpu_baseline'))[:900])"
echo "== host LAPACK probe (why is the cpu baseline 34 GFLOP/s?)"; date
timeout 200 python - <<'PY'
import time, numpy as np, scipy.linalg as sla, os
rng = np.random.default_rng(0); n = 8192
B = rng.normal(size=(n, 256)); K = B @ B.T + n * np.eye(n)
def t(label):
    t0 = time.perf_counter(); sla.cholesky(K, lower=True, check_finite=False); dt = time.perf_counter() - t0
    print(f"{label}: {dt:.2f} s = {(n**3/3)/dt/1e9:.0f} GFLOP/s", flush=True)
print("affinity", len(os.sched_getaffinity(0)), "OMP", os.environ.get("OMP_NUM_THREADS"), "OPENBLAS", os.environ.get("OPENBLAS_NUM_THREADS"))
t("default threads, before torch")
from threadpoolctl import threadpool_info, threadpool_limits
print([(d.get("internal_api"), d.get("num_threads"), d.get("threading_layer")) for d in threadpool_info()])
with threadpool_limits(limits=64): t("limit 64")
with threadpool_limits(limits=16): t("limit 16")
import torch
print("torch threads", torch.get_num_threads())
t("default threads, after torch")
print([(d.get("internal_api"), d.get("num_threads")) for d in threadpool_info()])
t0 = time.perf_counter(); A = K @ K; dt = time.perf_counter() - t0; print(f"dgemm {2*n**3/dt/1e9:.0f} GFLOP/s")
PY
echo "== kernel traces for the panel timeline"; date
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_c2_base -o bench -- python $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2>&1
TGP_HIP_OPTIONS="reserve_cus=8,trsm_split=1" timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_c2_protect -o bench -- python $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2>&1
cd $R
for d in prof_c2_base prof_c2_protect; do echo "-- $d"; python scripts/prof_top.py $(ls $O/$d/*.db | head -1) 12; done
date
} > $O/round.log 2>&1
tail -120 $O/round.log
