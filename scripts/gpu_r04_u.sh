#!/bin/bash
# pivot chain of the 4 x 4 diagonal blocks: one Newton step behind v_rcp_f64, products beside the reciprocal (probe builds)
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
for B in 512 64; do
  for rep in 1 2; do
    run $B X=0
    run $B SQPHIP_SO=scripts/probes/libsqphip_nr1.so
    run $B SQPHIP_SO=scripts/probes/libsqphip_fastpivot.so
  done
done
