#!/bin/bash
mkdir -p gpurun_out
for st in "$@"; do
  echo "=== $st ===" >> gpurun_out/diag.log
  timeout 75 python -u scripts/diag.py $st >> gpurun_out/diag.log 2>&1
  echo "exit $?" >> gpurun_out/diag.log
done
tail -150 gpurun_out/diag.log
