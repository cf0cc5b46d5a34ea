#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench workload: scripts/prof_bench.sh <outdir under gpurun_out>
# (run on the GPU box; writes <outdir>/prof/*.csv, <outdir>/timeline.csv, <outdir>/bench_under_rocprof.json)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out/prof
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra \
  > $out/bench_under_rocprof.json 2> $out/rocprof.err
cd $GRAFT_REPO_ROOT
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/timeline.py $t $out/timeline.csv
