#!/bin/bash
# Round profile set on the GPU box: scripts/r2_profile.sh <tag>   -> gpurun_out/<tag>/...
set -u
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
bash scripts/prof_bench.sh $tag > /dev/null
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- \
    python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2> $out/pmc_$c.err
done
python scripts/pmc_traffic.py $(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
  $(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $tag > $out/pmc.txt 2>&1
cp profiles/${tag}_pmc_traffic.json $out/ 2>/dev/null
timeout -s KILL 900 python bench.py > $out/bench.json 2> $out/bench.err
tail -c 3000 $out/bench.json
