#!/bin/bash
# A/B of ONE library under values of an environment knob, by rocprofv3 kernel stats: scripts/ab_env.sh <tag> <VAR> <v1> [v2 ...]
# (VC2_AB_LIB: the library, default the in-tree one; VC2_AB_WORKLOAD=long512 as in ab_stats.sh; VC2_AB_ARGS: extra bench.py arguments)
tag=$1; var=$2; shift; shift
lib=${VC2_AB_LIB:-vidcom2_amd/_lib/libvc2hip.so}
for v in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/${var}_$v; mkdir -p $out/prof
  if [ "${VC2_AB_WORKLOAD:-bench}" = "long512" ]; then cmd="python $GRAFT_REPO_ROOT/scripts/long512.py"; else cmd="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra ${VC2_AB_ARGS:-}"; fi
  ( cd /tmp; export TMPDIR=/tmp; export $var=$v; VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- $cmd > $out/bench.json 2> $out/rocprof.err )
  echo "== $var=$v: $(python -c "import json;print(json.load(open('$out/bench.json'))['ms_per_step'])" 2>/dev/null)"
  python scripts/kstats.py $out/prof
done
