#!/bin/bash
# round 3, batch 8: forward streaming solve with tf_b in LDS (pipelined hop) -- parity + timing
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b08
mkdir -p $O
export TMPDIR=/tmp
{
date
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -x -q -k "streaming or solver_protocol or predict or condition or n6144 or golden" 2>&1 | tail -5
timeout 200 python scripts/time_paths.py 16384 4096 2>&1 | tail -8
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps(d['roofline_secondary'])[:1500])"
timeout 200 python bench.py --workload n65536 --steps 1 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps(d['roofline_secondary'])[:1500])"
date
} > $O/log.txt 2>&1
cat $O/log.txt
