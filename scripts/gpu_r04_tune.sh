#!/bin/bash
run() { v=$(env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $B 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), d['config']['sweeps'], round(d['config']['factorisations_per_qp'],2))"); echo "batch $B $*: $v"; }
for B in 512 64; do
run X=0
run SQPHIP_MF_SPEC=1
run SQPHIP_MF_SPEC=2
run SQPHIP_MF_SPEC=0
run SQPHIP_TRANS_PERIOD=2
run SQPHIP_TRANS_PERIOD=4
run SQPHIP_TRANS_PERIOD=5
run SQPHIP_MF_SMALL_FRONT=24
run SQPHIP_MF_SMALL_FRONT=48
run SQPHIP_MF_ZERO_FRAC=0.15
run SQPHIP_MF_ZERO_FRAC=0.35
run SQPHIP_GROUPS=3
run SQPHIP_GROUPS=2
done
