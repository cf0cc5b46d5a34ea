for m in 0 1 2; do
  export VC2_RIDER_PROBE_MODE=$m
  out=gpurun_out/r2p_$m; mkdir -p $out/prof
  cd /tmp; export TMPDIR=/tmp
  VC2_LIB_PATH=$GRAFT_REPO_ROOT/vidcom2_amd/_lib/ab/probe.so timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  echo "mode $m: $(grep -h k_norm_colsum $out/prof/bench_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3)"
done
