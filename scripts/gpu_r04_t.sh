#!/bin/bash
# second shift per sweep in the tail of a run only (SQPHIP_MF_SPEC_TAIL = instances of a group with work left)
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
for B in 512 256; do
  for T in 0 16 32 48 64 96 0; do run $B SQPHIP_MF_SPEC_TAIL=$T; done
done
