#!/bin/bash
# QP/s of the two large shapes against the number of resident scenarios
mkdir -p gpurun_out
for cfg in "case1354 256" "case1354 512" "case1354 1024" "case9241 16" "case9241 64"; do
  set -- $cfg
  echo "== $1 batch $2" >> gpurun_out/big_cases.log
  timeout -k 10 420 python bench.py --workload $1 --batch $2 --no-cpu-baseline > gpurun_out/big_$1_$2.json 2>> gpurun_out/big_cases.log || { echo FAILED >> gpurun_out/big_cases.log; exit 1; }
  python scripts/print_bench.py gpurun_out/big_$1_$2.json >> gpurun_out/big_cases.log
done
