for l in vidcom2_amd/_lib/libvc2hip.so scripts/debug/libvc2hip_s1p1.so scripts/debug/libvc2hip_s1p2.so; do
  out=$GRAFT_REPO_ROOT/gpurun_out/s1probe/$(basename $l .so); mkdir -p $out/prof
  ( cd /tmp; export TMPDIR=/tmp; VC2_LIB_PATH=$GRAFT_REPO_ROOT/$l timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python $GRAFT_REPO_ROOT/scripts/dev/s1_only.py warm > $out/log 2>&1 )
  echo "== $l"; python scripts/kstats.py $out/prof | cut -c1-100
done
