# rocprofv3 kernel stats of a quick bench run.  usage: gpu_stats_quick.sh TAG [bench args]
TAG=${1:-q}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --quick "$@" > $O/bench.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
F=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp $F $R/gpurun_out/prof_${TAG}_kernel_stats.csv
python3 $R/scripts/trace_by_position.py $O/stats > $R/gpurun_out/prof_${TAG}_timeline.txt 2>&1
rm -rf $O/stats
cut -c1-160 $R/gpurun_out/prof_${TAG}_kernel_stats.csv | head -24
tail -1 $O/bench.json | cut -c1-120
