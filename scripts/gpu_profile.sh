# Profiles of bench.py for one workload (one GPU): kernel stats of the given command, then FETCH_SIZE and WRITE_SIZE in
# their own --pmc passes (no tracing flags with --pmc; the program itself after --).  Writes gpurun_out/prof_<tag>_<wl>/ ;
# scripts/make_profile.py turns it into the files committed under profiles/.
# usage: gpu_profile.sh TAG WORKLOAD [bench args for the stats pass]
TAG=${1:-r03}; WL=${2:-case118}; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_${TAG}_${WL}
rm -rf $O && mkdir -p $O
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --workload $WL "$@" > $O/bench_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
PMCARGS="--workload $WL --steps 2 --warmup 0 --quick --no-kernel-timing"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py $PMCARGS > $O/fetch.json 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py $PMCARGS > $O/write.json 2> $O/write.err || { tail -5 $O/write.err; exit 1; }
python3 $R/scripts/make_profile.py $O $TAG $WL
rm -rf $O/stats $O/fetch $O/write
