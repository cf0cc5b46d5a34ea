#!/bin/bash
# Profile set of the 512-frame clip (X = 686 MiB: beyond the 256 MiB Infinity Cache): scripts/prof_long512.sh <tag>
#   -> gpurun_out/<tag>_long512/{prof (kernel stats), pmc_FETCH_SIZE, pmc_WRITE_SIZE}, profiles/<tag>_long512_pmc_traffic.json
set -u
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_long512
mkdir -p $out/prof
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python $GRAFT_REPO_ROOT/scripts/long512.py > $out/long512.txt 2> $out/rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python $GRAFT_REPO_ROOT/scripts/long512.py > /dev/null 2> $out/pmc_$c.err
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_traffic.py $(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
  $(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) ${tag}_long512 "long512: 512x196x3584 bf16 r=0.25 (default 'torch' mode), one GPU" > $out/pmc.txt 2>&1
cp profiles/${tag}_long512_pmc_traffic.json $out/ 2>/dev/null
cat $out/long512.txt; python scripts/kstats.py $out/prof; cat $out/pmc.txt
