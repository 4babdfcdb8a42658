#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench (+ optional rocprof).  Outputs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== device ==" | tee gpurun_out/device.txt
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2) >> gpurun_out/device.txt 2>&1
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_X--x} --timeout 600 -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" | tee -a gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
echo "== bench =="
timeout 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 --stages > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit: $?"; cat gpurun_out/bench.log; tail -20 gpurun_out/bench.err
if [ "${ROCPROF:-1}" = "1" ]; then
echo "== rocprofv3 kernel-trace stats =="
cd /tmp 2>/dev/null; cd - >/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rocprof exit: $?"
tail -2 gpurun_out/prof_bench.log
find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
