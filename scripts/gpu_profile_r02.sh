# Round-2 profiles of bench.py (one GPU): kernel stats of the default command, then FETCH_SIZE and WRITE_SIZE in their
# own --pmc passes (no tracing flags with --pmc).  Writes gpurun_out/prof_<tag>/ ; scripts/make_profile_r02.py turns it
# into the files committed under profiles/.   usage: gpu_profile_r02.sh TAG [bench args for the stats pass]
TAG=${1:-r02}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 700 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py "$@" > $O/bench_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
PMCARGS="--steps 2 --warmup 0 --no-termination --no-cpu-baseline --no-dense-ldlt --no-screening --no-kernel-timing"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py $PMCARGS "$@" > $O/fetch.json 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py $PMCARGS "$@" > $O/write.json 2> $O/write.err || { tail -5 $O/write.err; exit 1; }
python3 $R/scripts/make_profile_r02.py $O $TAG
