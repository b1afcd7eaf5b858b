#!/bin/bash
run() { env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 0 --literal-quirks 0 --ipm-corrector $C --no-cpu-baseline --no-dense-ldlt --no-batch-curve --no-screening --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); t=d['termination']['literal_quirks_0']
print('corrector $C $*: termination leg', round(t['qp_per_s'],1), 'QP/s', round(t['seconds'],2), 's sweeps', t['sweeps'], 'qp', t['qp_solved'], 'conv', t['converged_ret0'], 'fac/qp', round(t['factorisations_per_qp'],2))"; }
for C in 1 0; do
run X=0
run SQPHIP_TRANS_PERIOD=1
run SQPHIP_MF_INERTIA_KERNEL=1
run SQPHIP_GROUPS=1
done
