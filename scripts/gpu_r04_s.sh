#!/bin/bash
# where the host threads of the instance groups spend their time (SQPHIP_HOST_STATS)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in 64 512; do
  echo "== batch $B"
  SQPHIP_HOST_STATS=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --quick --batch $B 2>&1 >/dev/null | grep "sqphip: group" | tail -4
done
echo "== batch 64, one group"
SQPHIP_GROUPS=1 SQPHIP_HOST_STATS=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --quick --batch 16 2>&1 >/dev/null | grep "sqphip: group" | tail -2
