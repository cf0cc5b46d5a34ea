#!/bin/bash
# SQ counter pass over one bench run: scripts/sq_pass.sh <tag> [lib.so] [counters...]  -> gpurun_out/<tag>/sq.txt
# (own rocprofv3 pass, --pmc with --kernel-trace only; median per dispatch and per kernel)
tag=$1; lib=${2:-vidcom2_amd/_lib/libvc2hip.so}; shift; shift
ctr=${@:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; rm -rf $out/sq; mkdir -p $out/sq
cd /tmp; export TMPDIR=/tmp
VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -s KILL 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/sq -o p -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2> $out/sq.err
cd $GRAFT_REPO_ROOT
python - "$out" <<'PY'
import csv, glob, re, statistics, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"\b(k_[a-z_0-9]+)", r["Kernel_Name"])
    if m: d[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, cs in d.items():
    med = {c: statistics.median(v) for c, v in cs.items()}
    wc = med.get("SQ_WAVE_CYCLES", 0) or 1
    lines.append(k + " " + " ".join(f"{c.replace('SQ_', '')}={v:.3g}({v / wc:.2f})" for c, v in sorted(med.items())))
open(out + "/sq.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
