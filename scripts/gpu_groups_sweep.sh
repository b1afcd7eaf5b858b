# QP/s of the quick bench against the number of instance groups (SQPHIP_GROUPS).  usage: gpu_groups_sweep.sh [bench args]
cd $GRAFT_REPO_ROOT
for g in 1 2 3 4 5 6 7; do
  SQPHIP_GROUPS=$g python bench.py --steps 20 --warmup 5 --quick "$@" 2>/dev/null | tail -1 > /tmp/g.json
  python -c "import json; d=json.load(open('/tmp/g.json')); print('groups', $g, round(d['value'],1), 'sweeps', d['config']['sweeps'])"
done
