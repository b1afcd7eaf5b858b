#!/bin/bash
# every instance group on its own quarter of the CUs (SQPHIP_CU_PARTITION = 1 contiguous mask bits, 2 every fourth bit)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # batch env...
  B=$1; shift
  out=$(env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --quick --batch $B 2>/dev/null | tail -1)
  python3 - "$B" "$*" "$out" <<'PY'
import json,sys
d=json.loads(sys.argv[3]); c=d["config"]
print(f"batch {sys.argv[1]} {sys.argv[2]}: {d['value']:.1f} sweeps {c['sweeps']} fac/qp {c['factorisations_per_qp']:.2f} qp {c['qp_solved']} fac {c['kkt_factorisations']}", flush=True)
PY
}
for B in 512 64 2048; do
  run $B SQPHIP_CU_PARTITION=0
  run $B SQPHIP_CU_PARTITION=1
  run $B SQPHIP_CU_PARTITION=2
done
