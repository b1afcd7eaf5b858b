#!/bin/bash
# L stored step by step inside the elimination of the static front kernels (default) against one burst at the end (probe build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "multifrontal or spine or kernel or front or sparse or factor" 2>&1 | tail -3
run() { # args... -- env...
  A=$1; shift
  out=$(env "$@" timeout -k 10 400 python bench.py --quick $A 2>/dev/null | tail -1)
  python3 - "$A" "$*" "$out" <<'PY'
import json,sys
d=json.loads(sys.argv[3]); c=d["config"]
print(f"{sys.argv[1]} {sys.argv[2]}: {d['value']:.1f} sweeps {c['sweeps']} fac/qp {c['factorisations_per_qp']:.2f} qp {c['qp_solved']} fac {c['kkt_factorisations']}", flush=True)
PY
}
P=SQPHIP_SO=scripts/probes/libsqphip_noearly.so
for A in "--steps 20 --warmup 5 --batch 512" "--steps 20 --warmup 5 --batch 64" "--workload case14" "--workload case1354"; do
  run "$A" X=0
  run "$A" $P
  run "$A" X=0
  run "$A" $P
done
