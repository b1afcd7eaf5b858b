#!/bin/bash
# idle time between the kernels of a queue, 64 and 512 resident scenarios (what a graph of the sweep could win)
R=${GRAFT_REPO_ROOT:-/root/repo}; F=$R/gpurun_out/gaps_r04; mkdir -p $F
cd /tmp && export TMPDIR=/tmp
for B in 64 512; do
  timeout -k 10 300 rocprofv3 --kernel-trace -d $F/trace$B --output-format csv -- python3 $R/scripts/gpu_sqp_run.py case118 $B 12 2 > $F/run$B.log 2> $F/trace$B.err || { tail -3 $F/trace$B.err; exit 1; }
  echo "== $B resident scenarios" >> $F/r04_launch_gaps.txt
  python3 $R/scripts/trace_gaps.py $F/trace$B 0.25 >> $F/r04_launch_gaps.txt
  rm -rf $F/trace$B
done
cat $F/r04_launch_gaps.txt
