for m in 0 1 2; do
  for wl in cfg2 target; do
  out=$GRAFT_REPO_ROOT/gpurun_out/r04_probe/m${m}_$wl; mkdir -p $out/prof
  ( cd /tmp; export TMPDIR=/tmp; VC2_RIDER_PROBE_MODE=$m VC2_LIB_PATH=$GRAFT_REPO_ROOT/scripts/lib_probe.so timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/err )
  echo "== mode $m $wl"; python $GRAFT_REPO_ROOT/scripts/kstats.py $out/prof | tail -1
  done
done
