#!/bin/bash
# per-instance fused solve kernel (k_mf_solve_inst) against the level-by-level launches at small batches
mkdir -p gpurun_out
run() { echo "== B=$B $*" >> gpurun_out/inst_solve.log; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-termination --no-dense-ldlt --batch ${B:-512} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['sweeps'], d['roofline']['frac'], d['roofline']['instance_groups'])" >> gpurun_out/inst_solve.log || exit 1; }
for B in 64 128 256; do
  export B
  run X=1
  run SQPHIP_MF_INST_SOLVE_MIN=1
  run SQPHIP_GROUPS=1
  run SQPHIP_GROUPS=1 SQPHIP_MF_INST_SOLVE_MIN=1
done
