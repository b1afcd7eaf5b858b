#!/bin/bash
# round 4: spine kernel on / off against the number of instance groups and the transition period (quick benches, 4 s each)
mkdir -p gpurun_out
out=gpurun_out/r04_c_spine_groups.txt
: > $out
for b in 512 64; do
for sp in 1 0; do
for g in 2 3 4; do
  v=$(SQPHIP_MF_SPINE=$sp SQPHIP_GROUPS=$g timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $b 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['roofline']['factor_seconds'],3), round(d['roofline']['solve_seconds'],3))")
  echo "batch $b spine $sp groups $g: $v" | tee -a $out
done; done; done
