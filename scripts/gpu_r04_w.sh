#!/bin/bash
# (a) lane swaps / readlane instead of shuffles / LDS in the 4 x 4 blocks: same work counters as before (660 924 at 512, 80 755 at 64)?
# (b) static front kernels from T tile rows on (SQPHIP_MF_STATIC_MIN): the generic kernel of the smallest fronts runs at one wave per SIMD
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # args... -- env...
  A=$1; shift
  out=$(env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --quick $A 2>/dev/null | tail -1)
  python3 - "$A" "$*" "$out" <<'PY'
import json,sys
d=json.loads(sys.argv[3]); c=d["config"]
print(f"{sys.argv[1]} {sys.argv[2]}: {d['value']:.1f} sweeps {c['sweeps']} fac/qp {c['factorisations_per_qp']:.2f} qp {c['qp_solved']} fac {c['kkt_factorisations']}", flush=True)
PY
}
for A in "--batch 512" "--batch 64"; do
  for M in 4 3 2 1 4; do run "$A" SQPHIP_MF_STATIC_MIN=$M; done
done
for M in 4 1; do
  out=$(env SQPHIP_MF_STATIC_MIN=$M timeout -k 10 300 python bench.py --workload case14 --quick 2>/dev/null | tail -1); python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('case14 static_min $M:', round(d['value'],1))" "$out"
done
