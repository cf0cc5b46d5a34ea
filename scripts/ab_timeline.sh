#!/bin/bash
# A/B of library variants by one-pass rocprofv3 timelines: scripts/ab_timeline.sh <tag> <lib1.so> [lib2.so ...]
# (each variant: rocprofv3 --kernel-trace of bench.py --steps 20 --warmup 5, then scripts/timeline.py)
tag=$1; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/$name; mkdir -p $out/prof
  ( cd /tmp; export TMPDIR=/tmp; VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/rocprof.err )
  t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
  echo "== $name: $(python -c "import json;print(json.load(open('$out/bench.json'))['ms_per_step'])")"
  [ -n "$t" ] && python scripts/timeline.py $t $out/timeline.csv > /dev/null; python scripts/kstats.py $out/prof
done
