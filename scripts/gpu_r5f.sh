#!/bin/bash
# round 5, batch f: left-looking forward solve and the default follower (poll kernel again) on the GPU; timings of the
# block-column path's new operations at world size 1.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5f}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== pytest: distributed + variants + recovery"
timeout 1200 python -m pytest tests/test_gpu_5_distributed.py tests/test_gpu_6_multirank_one_gpu.py tests/test_gpu_1_gp.py tests/test_gpu_0_kernels.py -x -q -m gpu -p no:cacheprovider --durations=5 -k "gradient or resident or peers or timed_out or variants or rccl or config2 or stress" 2>&1 | tail -14
echo "== timings, N = 16384"; date
timeout 600 python scripts/dist_timing.py 16384 1024 2>&1 | tail -12
echo "== timings, N = 65536"; date
timeout 900 python scripts/dist_timing.py 65536 1024 2>&1 | tail -12
echo "== bench default"; date
timeout 600 python bench.py --no-north-star --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
date
} > $O/log.txt 2>&1
tail -80 $O/log.txt | cut -c1-300
