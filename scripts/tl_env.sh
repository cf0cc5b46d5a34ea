#!/bin/bash
# One-pass timelines of a workload under values of an environment knob: scripts/tl_env.sh <tag> <workload> <VAR> <v1> [v2 ...]
# (VC2_AB_LIB: the library, default the in-tree one).  Prints one line per value: kernel durations in launch order.
tag=$1; wl=$2; var=$3; shift; shift; shift
lib=${VC2_AB_LIB:-vidcom2_amd/_lib/libvc2hip.so}
for v in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/${wl}_${var}_$v; mkdir -p $out/prof
  ( cd /tmp; export TMPDIR=/tmp; export $var=$v; VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/rocprof.err )
  t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python scripts/timeline.py $t $out/timeline.csv > /dev/null
  echo "$wl $var=$v: $(python - "$out/timeline.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(" ".join(f"{r['kernel'][2:]} {r['dur_us']}" for r in rows if r['kernel'] != 'TOTAL'), "| TOTAL", rows[-1]['dur_us'])
PY
)"
done
