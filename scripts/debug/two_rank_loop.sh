#!/bin/bash
# Repeat bench.py's two-ranks-on-one-GPU run (gloo) to catch an intermittent hang: scripts/debug/two_rank_loop.sh <n> <workload>
# Every run has a watchdog (VC2_BENCH_WATCHDOG: Python stacks of all threads on expiry) and a hard timeout.
n=${1:-10}; wl=${2:-cfg4}
out=$GRAFT_REPO_ROOT/gpurun_out/two_rank; mkdir -p $out
for i in $(seq 1 $n); do
  port=$((20000 + RANDOM % 20000))
  S=$(date +%s)
  VC2_BENCH_ONE_GPU=1 VC2_BENCH_BACKEND=gloo VC2_BENCH_CPU_THREADS=16,32 VC2_BENCH_WATCHDOG=150 timeout -s KILL 240 \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus 2 --steps 2 --warmup 1 --workload $wl > $out/run_$i.out 2> $out/run_$i.err
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - S )) s $(grep -c '^{' $out/run_$i.out) line(s)"
  if [ $rc -ne 0 ]; then echo "---- stderr tail of run $i"; tail -60 $out/run_$i.err; fi
done
