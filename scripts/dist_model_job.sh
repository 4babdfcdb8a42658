cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6o; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d /tmp/dtl -o bench -- python bench.py --distributed --workload c4 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-profile > $O/c4_line.json 2>$O/err.txt
python scripts/timeline.py $(ls /tmp/dtl/*.db | head -1) /tmp/dtl.csv 20000 > /dev/null; cp /tmp/dtl.csv $O/c4_world1_timeline.csv
for link in 120 60 40; do echo "## xGMI at $link GB/s effective per link"; python scripts/dist_model.py /tmp/dtl.csv 131072 1024 $link 66; done > $O/dist_model_c4.txt 2>&1
tail -c 600 $O/c4_line.json; head -60 $O/dist_model_c4.txt
