#!/bin/bash
# static front kernels for nine to twelve tile rows with the larger unroll budget: kernel tests, work counters of the headline (must be
# 660 924 / 6 140 sweeps), the two large shapes against the generic kernel (SQPHIP_MF_STATIC_MAX=8)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "multifrontal or kernel or front or sparse or factor or case1354 or case9241" 2>&1 | tail -3
run() { # args... -- env...
  A=$1; shift
  out=$(env "$@" timeout -k 10 600 python bench.py --quick $A 2>/dev/null | tail -1)
  python3 - "$A" "$*" "$out" <<'PY'
import json,sys
d=json.loads(sys.argv[3]); c=d["config"]
print(f"{sys.argv[1]} {sys.argv[2]}: {d['value']:.2f} sweeps {c['sweeps']} fac/qp {c['factorisations_per_qp']:.2f} qp {c['qp_solved']} fac {c['kkt_factorisations']}", flush=True)
PY
}
run "--steps 20 --warmup 5 --batch 512" X=0
run "--workload case1354" X=0
run "--workload case1354" SQPHIP_MF_STATIC_MAX=8
run "--workload case9241" X=0
