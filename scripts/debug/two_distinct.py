"""Enqueue-only throughput with k plans on k streams, DISTINCT clips vs the same clip."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidcom2_amd import synth
from vidcom2_amd.vidcom2 import CompressPlan
F, N, D, dt = 128, 196, 3584, torch.bfloat16
xs = [synth.make(F, N, D, dt, s, "drift").cuda() for s in range(4)]
for k in (1, 2, 4):
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(k - 1)]
    plans = []
    for st in streams:
        with torch.cuda.stream(st):
            plans.append(CompressPlan(F, N, D, dt, xs[0].device, 0.25))
    for name, pick in (("same clip", lambda i: xs[0]), ("distinct clips", lambda i: xs[i % 4])):
        def rnd():
            for i, (st, pl) in enumerate(zip(streams, plans)):
                with torch.cuda.stream(st):
                    pl.enqueue(pick(i))
        for _ in range(20): rnd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): rnd()
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / 100
        print(f"{k} in flight, {name}: {dt_s / k * 1e6:.1f} us per clip, {k * F * N / dt_s / 1e6:.1f} M tokens/s", flush=True)
