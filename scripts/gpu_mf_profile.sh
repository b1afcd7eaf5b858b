# rocprofv3 kernel stats of a batched SQP run through the sparse path.  Usage: gpu_mf_profile.sh TAG CASE BATCH STEPS [MODE]
TAG=${1:-mf0}; CASE=${2:-case118}; BATCH=${3:-64}; STEPS=${4:-6}; MODE=${5:-2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/scripts/gpu_sqp_run.py $CASE $BATCH $STEPS $MODE > $O/run.log 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
cat $O/run.log
F=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp $F $R/gpurun_out/prof_${TAG}_kernel_stats.csv
head -40 $F
