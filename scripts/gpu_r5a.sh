#!/bin/bash
# round 5, batch a: the driver's own GPU check on the current tree -- the full suite IN DRIVER ORDER with -x -q --,
# smoke, the default bench line, and two probes for the poller work (stream wait-value; serialised kernels).
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5a}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== pytest tests -x -q -m gpu (driver order)"
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=8 2>&1 | tail -25
echo "== smoke"; date
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench default (the driver's line)"; date
timeout 600 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-900
echo "== probe: stream wait-value"; date
timeout 60 scripts/probe_waitvalue 2>&1 | tail -20
echo "== default path under AMD_SERIALIZE_KERNEL=3 (n4096, n16384)"; date
for n in n4096 c2; do AMD_SERIALIZE_KERNEL=3 timeout 120 python bench.py --no-cpu-baseline --no-secondary --no-north-star --no-profile --workload $n --steps 3 --warmup 1 2>&1 | tail -2 | cut -c1-300; done
date
} > $O/log.txt 2>&1
tail -80 $O/log.txt | cut -c1-600
