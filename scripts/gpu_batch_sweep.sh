#!/bin/bash
# QP/s of one GPU against the number of resident scenarios: predicts the strong-scaling curve of 512 scenarios over N GPUs
mkdir -p gpurun_out
for b in 512 256 128 64; do
  for g in ${GROUPS_LIST:-default}; do
    echo "== batch $b groups $g" >> gpurun_out/batch_sweep.log
    if [ "$g" = default ]; then unset SQPHIP_GROUPS; else export SQPHIP_GROUPS=$g; fi
    timeout -k 10 200 python bench.py --batch $b --no-cpu-baseline --no-termination --no-dense-ldlt ${EXTRA} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['sweeps'], d['roofline']['frac'], d['roofline']['instance_groups'])" >> gpurun_out/batch_sweep.log || exit 1
  done
done
