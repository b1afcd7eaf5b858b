#!/bin/bash
run() { v=$(env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $B 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), d['config']['sweeps'], round(d['config']['factorisations_per_qp'],2))"); echo "batch $B $*: $v"; }
for B in 256 128; do
run SQPHIP_MF_SPEC=1
run SQPHIP_MF_SPEC=2
run SQPHIP_MF_SPEC=0
done
B=1024; run X=0
B=2048; run X=0
