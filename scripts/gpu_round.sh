#!/bin/bash
# round 2, session 2, batch 4: persistent trailing update with reserved workgroup slots beside a chain
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/s2b4.log
: > $L
echo "== pytest -m gpu (kernels, gp)" >> $L; date >> $L
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -x -q 2>&1 | tail -4 >> $L
echo "== gemm alone" >> $L
timeout 300 python scripts/gemm_bench.py f64 16384 2>&1 | grep -E "K= *(512|1024|2048)" >> $L
echo "== bench lines: fused_step x chain_reserve" >> $L; date >> $L
for w in c2 n8192 n32768; do
  for f in 0 1; do
    for r in 0 32 64 128; do
      echo "# $w fused_step=$f chain_reserve=$r" >> $L
      TGP_HIP_OPTIONS="fused_step=$f,chain_reserve=$r" timeout 600 python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline') or {}
    print(json.dumps({'evals_s':round(d['value'],3),'ms':round(d['ms_per_step'],3),'syrk_TF':round(r.get('achieved',0),2),'chol_TF':round(d.get('cholesky_tflops',0),2)}))
" >> $L
    done
  done
done
echo "# n65536 chain_reserve=0 / 64" >> $L
for r in 0 64; do
TGP_HIP_OPTIONS="fused_step=0,chain_reserve=$r" timeout 600 python bench.py --workload n65536 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-300 >> $L
done
date >> $L
tail -70 $L
