# Profiles of the default bench.py run (one GPU): kernel stats, then FETCH_SIZE and WRITE_SIZE in their own
# --pmc passes (no tracing flags with --pmc).  Writes gpurun_out/prof_<tag>/ and summary JSON/CSV next to it.
TAG=${1:-r01c}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/fetch.json 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/write.json 2> $O/write.err || { tail -5 $O/write.err; exit 1; }
python3 $R/scripts/make_traffic_profile.py $O $TAG
