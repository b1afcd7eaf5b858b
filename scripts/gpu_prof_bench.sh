# rocprofv3 kernel stats of the bench's timed region (no extra legs).  usage: gpu_prof_bench.sh TAG [ENV=val ...] -- [bench args]
TAG=$1; shift
ENVS=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS="$ENVS $1"; shift; done
[ "$1" = "--" ] && shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in $ENVS; do export $e; done
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --no-termination --no-cpu-baseline --no-dense-ldlt "$@" > $O/bench.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
F=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp $F $R/gpurun_out/prof_${TAG}_kernel_stats.csv
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('$TAG', round(d['value']), 'QP/s', round(d['ms_per_step'],1), 'ms/step')"
head -16 $F | cut -c1-150
