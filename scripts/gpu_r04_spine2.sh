#!/bin/bash
for b in 512 256 128 64; do
for sp in 0 1; do
  v=$(SQPHIP_MF_SPINE=$sp timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $b 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), d['config']['sweeps'])")
  echo "batch $b spine $sp: $v"
done; done
