#!/bin/bash
# One-pass timeline + kernel stats of another bench workload: scripts/prof_workload.sh <tag> <workload> (cfg2, cfg3, ...)
tag=$1; wl=$2
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_$wl; mkdir -p $out/prof
( cd /tmp; export TMPDIR=/tmp; timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/rocprof.err )
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python scripts/timeline.py $t $out/timeline.csv
python scripts/kstats.py $out/prof
