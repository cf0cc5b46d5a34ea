#!/bin/bash
# A/B of library variants by rocprofv3 kernel stats (averages over ~1500 launches each): scripts/ab_stats.sh <tag> <lib.so> ...
# VC2_AB_WORKLOAD=long512 profiles scripts/long512.py (a 512-frame clip: beyond the Infinity Cache) instead of bench.py
tag=$1; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/$name; mkdir -p $out/prof
  if [ "${VC2_AB_WORKLOAD:-bench}" = "long512" ]; then cmd="python $GRAFT_REPO_ROOT/scripts/long512.py"; else cmd="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"; fi
  ( cd /tmp; export TMPDIR=/tmp; VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- $cmd > $out/bench.json 2> $out/rocprof.err )
  echo "== $name: $(tail -c 300 $out/bench.json | head -c 300)"
  python scripts/kstats.py $out/prof
done
