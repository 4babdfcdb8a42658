#!/bin/bash
# round 3, batch 28: schedule options re-swept on the final kernels (c2 and N = 8 192 / 32 768)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b28
mkdir -p $O
{
date
timeout 900 python scripts/sweep.py 16384 9 "" "nb_outer=768" "nb_outer=1280" "nb_outer=1536" "first_split=4" "first_split=6" "first_split=7" "first_small_tiles=600" "first_small_tiles=2200" "first_small_tiles=0" "chain_reserve=64" "chain_reserve=192" "reserve_max_tiles=2400" "reserve_max_tiles=600" "sub_panel=512" "nb_first=512" "" 
timeout 600 python scripts/sweep.py 8192,32768 7 "" "nb_outer=768" "nb_outer=1536" "first_small_tiles=2200" "first_split=6" ""
date
} > $O/log.txt 2>&1
cat $O/log.txt
