#!/bin/bash
# hunt (5): after the fix (potf2 text included at kernel scope) -- default and fused chains, lookahead 0 / 1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/s2b15.log
: > $L
for o in "lookahead=0" "lookahead=1" "fused_step=1" "fused_step=1 lookahead=0"; do
  echo "## today (fixed) $o" >> $L
  timeout 150 python scripts/stress_nan.py 3000 25000 $o 2>&1 | grep -v "Warning\|msg +=" | cut -c1-100 >> $L
done
timeout 100 python scripts/stress_nan.py 5000 6000 lookahead=0 2>&1 | grep -v "Warning\|msg +=" | cut -c1-100 >> $L
date >> $L
cat $L
