#!/bin/bash
# QP/s of the default bench against the number of instance groups and HIP hardware queues
mkdir -p gpurun_out
echo "nproc $(nproc)" >> gpurun_out/groups_queues.log
for cfg in ${CFGS:-"4 4" "5 8" "6 8" "6 16" "7 16"}; do
  set -- $cfg
  echo "== groups $1 hw queues $2" >> gpurun_out/groups_queues.log
  SQPHIP_GROUPS=$1 GPU_MAX_HW_QUEUES=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-termination --no-dense-ldlt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['sweeps'], d['roofline']['frac'])" >> gpurun_out/groups_queues.log || exit 1
done
