#!/usr/bin/env python
"""compress_batch call by call (24 same-shape clips, results dropped / kept): python scripts/dev/batch_calls.py [f16|bf16] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidcom2_amd import synth, vidcom2 as V
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else torch.bfloat16
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3584
F, N = 128, 196
clips = [synth.make(F, N, D, dt, sd, "drift").cuda() for sd in range(4)]
batch = [clips[i % 4] for i in range(24)]
for keep in (False, True):
    for k in (1, 2, 3, 4):
        ts = []
        held = []
        for i in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = V.compress_batch(batch, N, 0.25, in_flight=k)
            ts.append((time.perf_counter() - t0) * 1e6 / 24)
            if keep: held = r
            del r
        print(f"keep={keep} in_flight={k}: " + " ".join(f"{t:.0f}" for t in ts), flush=True)
