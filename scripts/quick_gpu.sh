#!/bin/bash
# Quick GPU check of a kernel change: scripts/quick_gpu.sh <tag> [pytest -k expression]
# selection / full-pass parity tests, in-kernel stamps (debug library, if built), one-pass timeline from a rocprofv3
# kernel trace (the reliable per-kernel durations), a short bench line
tag=$1; kexpr=${2:-"selection_kat or full_pass or stage_functions or select_with_tpf or many_frames or loop_bounds or many_short"}
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$kexpr" > $out/gpu_tests.txt 2>&1; tail -4 $out/gpu_tests.txt
if [ -f scripts/libvc2hip_dbg.so ]; then python scripts/dbg_timing.py scripts/libvc2hip_dbg.so > $out/stamps.txt 2>&1; grep -v amdgpu.ids $out/stamps.txt | head -${STAMP_LINES:-60}; fi
bash scripts/prof_bench.sh $tag > /dev/null 2>&1; cat $out/timeline.csv
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $out/bench_quick.json 2> $out/bench.err
python -c "
import json; d=json.load(open('$out/bench_quick.json')); print(d['ms_per_step'], d['kernels_us'])"
