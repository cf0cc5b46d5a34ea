#!/bin/bash
# evaluate a library: timeline + replay counts + the soak's six `cancel` misses + adversarial fixtures in the default mode
lib=$1
export PYTHONPATH=$GRAFT_REPO_ROOT
bash scripts/ab_tl.sh $lib | tail -1 | sed 's/k_chan_stats.*k_norm_fix=[0-9.]* //; s/k_dist.*TOTAL/TOTAL/'
VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib python scripts/flag_counts.py drift | head -1
VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib python scripts/flag_counts.py iid | head -1
for s in 3319 3355 4716 6832 7196 8441; do VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib python tests/tools/soak_shapes_gpu.py $s $((s+1)) 2>&1 | tail -1 | cut -c1-60; done
VC2_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_long_clips.py -q -k "adversarial or long_clip_matches" 2>&1 | tail -3
