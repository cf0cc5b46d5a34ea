#!/usr/bin/env python
"""Clips in flight on k streams (one plan per stream, no allocation): us per clip for k = 1 .. 4.  python scripts/dev/lanes.py [bf16|f16] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidcom2_amd import synth, vidcom2 as V
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else torch.bfloat16
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3584
F, N = 128, 196
clips = [synth.make(F, N, D, dt, sd, "drift").cuda() for sd in range(4)]
batch = [clips[i % 4] for i in range(24)]
def T(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for k in (1, 2, 3, 4):
    cur = torch.cuda.current_stream()
    streams = [cur] + [torch.cuda.Stream() for _ in range(k - 1)]
    plans = []
    for s in streams:
        with torch.cuda.stream(s):
            plans.append(V.CompressPlan(F, N, D, dt, clips[0].device, 0.25))
    def run():
        for s in streams[1:]: s.wait_stream(cur)
        for i, b in enumerate(batch):
            plans[i % k].enqueue(b, stream=streams[i % k])
        for s in streams[1:]: cur.wait_stream(s)
    for _ in range(2): run()
    print(f"{k} clip(s) in flight: {T(run, 5) * 1e6 / len(batch):.1f} us per clip", flush=True)
for k in (1, 2, 3, 4):
    for _ in range(2): V.compress_batch(batch, N, 0.25, in_flight=k)
    print(f"compress_batch(in_flight={k}): {T(lambda: V.compress_batch(batch, N, 0.25, in_flight=k), 5) * 1e6 / len(batch):.1f} us per clip", flush=True)
distinct = [synth.make(F, N, D, dt, 100 + sd, "drift").cuda() for sd in range(16)]
for k in (2, 3):
    for _ in range(2): V.compress_batch(distinct, N, 0.25, in_flight=k)
    print(f"compress_batch(16 distinct clips, in_flight={k}): {T(lambda: V.compress_batch(distinct, N, 0.25, in_flight=k), 5) * 1e6 / 16:.1f} us per clip", flush=True)
