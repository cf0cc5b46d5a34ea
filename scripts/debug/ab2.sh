#!/bin/bash
# A/B over library builds: scripts/ab2.sh lib1.so lib2.so ...   (each run: bench without the CPU baseline)
for lib in "$@"; do
  VC2_LIB_PATH=$PWD/$lib python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_us']
print('$lib', '| ms', d['ms_per_step'], '| cfg2', d.get('cfg2',{}).get('ms_per_step'), '| exact', d.get('exact_mode',{}).get('ms_per_step'), '|', {n:k[n] for n in k})"
done
