#!/bin/bash
# A/B helper: run bench with env settings given as args "K=V K=V" per line of stdin
while read -r line; do
  [ -z "$line" ] && continue
  env $line python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_us']
print('$line', '| ms', d['ms_per_step'], '| norm', k.get('k_norm_colsum'), 'dist', k.get('k_dist'), 'chansel', k.get('k_chan_select'), 'sel', k.get('k_select'))"
done
