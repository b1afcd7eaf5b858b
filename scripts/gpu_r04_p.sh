#!/bin/bash
run() { v=$(env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $B 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), d['config']['sweeps'], round(d['config']['factorisations_per_qp'],2))"); echo "batch $B $*: $v"; }
for B in 512 256 64; do
run X=0
run SQPHIP_SO=scripts/probes/libsqphip_t512.so
run X=0
run SQPHIP_SO=scripts/probes/libsqphip_t512.so
done
