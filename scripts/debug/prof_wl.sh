#!/bin/bash
# rocprofv3 one-pass timeline of another bench workload: scripts/prof_wl.sh <workload> <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$2; mkdir -p $out/prof
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/err.txt
cd $GRAFT_REPO_ROOT
python scripts/timeline.py $(find $out/prof -name "*kernel_trace.csv" | head -1) $out/timeline.csv
