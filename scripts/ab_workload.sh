#!/bin/bash
# A/B of library builds on another bench workload: scripts/ab_workload.sh <tag> <workload> <lib.so> ...
tag=$1; wl=$2; shift; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/${name}_$wl; mkdir -p $out/prof
  ( cd /tmp; export TMPDIR=/tmp; VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/rocprof.err )
  echo "== $name $wl: $(python -c "import json;print(json.load(open('$out/bench.json'))['ms_per_step'])")"
  python scripts/kstats.py $out/prof
done
