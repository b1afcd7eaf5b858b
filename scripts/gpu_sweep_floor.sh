#!/bin/bash
# (1) pipelined vs lock-step counter read; (2) wall time of a fully gated-off sweep (launch-chain floor)
mkdir -p gpurun_out
run() { echo "== $*" >> gpurun_out/sweep_floor.log; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-termination --no-dense-ldlt --batch ${B:-512} 2>> gpurun_out/sweep_floor.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['sweeps'], d['roofline']['frac'], d['roofline']['instance_groups'])" >> gpurun_out/sweep_floor.log || exit 1; }
run X=1
run SQPHIP_SWEEP_LOCKSTEP=1
run SQPHIP_GROUPS=1
run SQPHIP_GROUPS=1 SQPHIP_SWEEP_LOCKSTEP=1
run SQPHIP_GROUPS=1 SQPHIP_EMPTY_SWEEPS=200
B=64 run SQPHIP_GROUPS=1
B=64 run SQPHIP_GROUPS=1 SQPHIP_SWEEP_LOCKSTEP=1
B=64 run SQPHIP_GROUPS=1 SQPHIP_EMPTY_SWEEPS=200
