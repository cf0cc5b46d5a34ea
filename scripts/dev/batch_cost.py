#!/usr/bin/env python
"""Where compress_batch's time goes (cfg5: 16 clips x 128x196x4096 fp16): host time of the enqueue loop, GPU time with 1 / 2 / 3
lanes, cProfile of one batch; and vc2_keep_positions: kernel launch + counts vs the Python around it."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidcom2_amd import synth, vidcom2 as V
from vidcom2_amd.fused import keep_positions
F, N, D, dt = 128, 196, 4096, torch.float16
clips4 = [synth.make(F, N, D, dt, sd, "drift").cuda() for sd in range(4)]
batch = [clips4[i % 4] for i in range(16)]
def T(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for lanes in (1, 2, 3, 4):
    for _ in range(2): V.compress_batch(batch, N, 0.25, in_flight=lanes)
    print(f"compress_batch, {lanes} lane(s): {T(lambda: V.compress_batch(batch, N, 0.25, in_flight=lanes), 5) * 1e3:.3f} ms per batch of 16")
plan = V.CompressPlan(F, N, D, dt, clips4[0].device, 0.25)
for _ in range(3): plan.enqueue(clips4[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(16): plan.enqueue(batch[i])
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"16 enqueues, one plan, one stream: host returns after {th * 1e6 / 16:.1f} us per clip; GPU {T(lambda: [plan.enqueue(b) for b in batch], 3) * 1e6 / 16:.1f} us per clip")
s2 = torch.cuda.Stream()
plans = [plan, None]
with torch.cuda.stream(s2):
    plans[1] = V.CompressPlan(F, N, D, dt, clips4[0].device, 0.25)
def two_lanes():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    for i, b in enumerate(batch):
        plans[i & 1].enqueue(b, stream=(cur, s2)[i & 1])
    cur.wait_stream(s2)
for _ in range(2): two_lanes()
print(f"16 enqueues, two plans, two streams (no allocation at all): {T(two_lanes, 5) * 1e6 / 16:.1f} us per clip")
pr = cProfile.Profile(); pr.enable(); V.compress_batch(batch, N, 0.25, in_flight=2); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
# keep_positions
dev = clips4[0].device
nvid, ntext = 64 * 324, 96
vm = torch.zeros(nvid + ntext, dtype=torch.bool, device=dev); vm[32:32 + nvid] = True
kept = torch.arange(0, nvid, 8, device=dev, dtype=torch.int64)
for _ in range(3): keep_positions(vm, kept, nvid)
print(f"keep_positions (python call): {T(lambda: keep_positions(vm, kept, nvid), 50) * 1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): keep_positions(vm, kept, nvid)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
