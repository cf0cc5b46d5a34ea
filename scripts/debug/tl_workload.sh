#!/bin/bash
# one-pass timeline of a bench workload: scripts/tl_workload.sh <workload> [lib.so]
w=$1; lib=${2:-vidcom2_amd/_lib/libvc2hip.so}
out=$GRAFT_REPO_ROOT/gpurun_out/tlw; rm -rf $out; mkdir -p $out/prof
cd /tmp; export TMPDIR=/tmp
VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extra > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
echo "== $w"; python scripts/timeline.py $(find $out/prof -name "*kernel_trace.csv" | head -1) | awk -F, '{printf "%s=%s ", $1, $4} END {print ""}'
