#!/bin/bash
# one-pass timelines for several library builds: scripts/ab_tl.sh lib1.so lib2.so ...
for lib in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/abtl; rm -rf $out; mkdir -p $out/prof
  cd /tmp; export TMPDIR=/tmp
  VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  echo "== $lib"; python scripts/timeline.py $(find $out/prof -name "*kernel_trace.csv" | head -1) | awk -F, '{printf "%s=%s ", $1, $4} END {print ""}'
done
