#!/bin/bash
# round 3, batch 4: 64x64-tile fp64 GEMM with LDS-direct 3-stage staging -- parity suite, then timings
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b04
mkdir -p $O
export TMPDIR=/tmp
{
date
timeout 900 python -m pytest tests -m gpu -x -q -k "not n131072 and not n262144 and not m4096" 2>&1 | tail -12
date
} > $O/pytest.log 2>&1
{
date
timeout 300 python scripts/sweep.py 16384 "" "sub_panel=512" "first_small_tiles=2000" "first_small_tiles=4000" "first_split=6" "nb_outer=768" "chain_reserve=0" ""
timeout 200 python scripts/sweep.py 8192,4096,2048 "" "sub_panel=512" ""
timeout 200 python scripts/sweep.py 32768 5 "" "sub_panel=512"
date
} > $O/sweep.log 2>&1
cat $O/pytest.log $O/sweep.log
