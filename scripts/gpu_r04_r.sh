#!/bin/bash
# more instance groups than four with more hardware queues
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
  run $B X=0
  run $B GPU_MAX_HW_QUEUES=8
  run $B GPU_MAX_HW_QUEUES=8 SQPHIP_GROUPS=5
  run $B GPU_MAX_HW_QUEUES=8 SQPHIP_GROUPS=6
  run $B GPU_MAX_HW_QUEUES=8 SQPHIP_GROUPS=7
  run $B SQPHIP_GROUPS=6
  run $B SQPHIP_SIDE_TRANS=1
  run $B GPU_MAX_HW_QUEUES=8 SQPHIP_SIDE_TRANS=1
  run $B GPU_MAX_HW_QUEUES=8 SQPHIP_SIDE_TRANS=1 SQPHIP_GROUPS=3
done
