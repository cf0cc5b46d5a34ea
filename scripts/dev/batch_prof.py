#!/usr/bin/env python
"""cProfile of compress_batch (24 same-shape clips): python scripts/dev/batch_prof.py [f16|bf16] [D] [lanes]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidcom2_amd import synth, vidcom2 as V
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else torch.bfloat16
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3584
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
F, N = 128, 196
clips = [synth.make(F, N, D, dt, sd, "drift").cuda() for sd in range(4)]
batch = [clips[i % 4] for i in range(24)]
for _ in range(3): V.compress_batch(batch, N, 0.25, in_flight=k)
torch.cuda.synchronize()
t0 = time.perf_counter(); r = V.compress_batch(batch, N, 0.25, in_flight=k); t1 = time.perf_counter()
print(f"one call: {(t1 - t0) * 1e6:.0f} us = {(t1 - t0) * 1e6 / 24:.1f} per clip")
pr = cProfile.Profile(); pr.enable(); V.compress_batch(batch, N, 0.25, in_flight=k); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
