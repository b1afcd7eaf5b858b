#!/bin/bash
# kernel stats of one bench command under rocprofv3 (--kernel-trace --stats only): gpu_r04_stats.sh TAG <bench args>
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/stats_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/raw --output-format csv -- python3 $R/bench.py "$@" > $O/bench_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
f=$(find $O/raw -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
rm -rf $O/raw
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:28]:
    print(f"{float(r['Percentage']):6.2f} %  {int(r['Calls']):7d} calls  {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:70]}")
PY
