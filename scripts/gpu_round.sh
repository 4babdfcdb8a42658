#!/bin/bash
# round 2, session 2, batch 3: kernel timelines of the fused and the unfused panel chain (c2, N = 4096)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
mkdir -p $O
L=$O/s2b3.log
: > $L
B="--no-cpu-baseline --no-secondary"
cd /tmp; export TMPDIR=/tmp
for f in 1 0; do
  for w in c2 n4096; do
    rm -rf $O/prof_${w}_f$f
    TGP_HIP_OPTIONS="fused_step=$f" timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_${w}_f$f -o bench -- python $R/bench.py --workload $w --steps 3 --warmup 1 $B > /dev/null 2>&1
    echo "== $w fused_step=$f" >> $L
    python $R/scripts/prof_top.py $(ls $O/prof_${w}_f$f/*.db | head -1) 10 >> $L 2>&1
    python $R/scripts/timeline.py $(ls $O/prof_${w}_f$f/*.db | head -1) /tmp/tl.csv 2500 > /dev/null
    if [ $w = c2 ]; then python $R/scripts/timeline_panels.py /tmp/tl.csv dump 9 1 >> $L 2>&1; else python $R/scripts/timeline_panels.py /tmp/tl.csv dump 2 1 >> $L 2>&1; fi
    rm -rf $O/prof_${w}_f$f
  done
done
date >> $L
tail -5 $L
