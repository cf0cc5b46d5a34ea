"""Throughput when several clips are compressed concurrently, one stream per clip (serving / batched eval)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidcom2_amd import synth, _ffi
import vidcom2_amd as vc
F, N, D = 128, 196, 3584
dev = torch.device("cuda:0")
xs = [synth.make(F, N, D, torch.bfloat16, s).to(dev) for s in range(4)]
for mode in ("torch", "exact"):
    _ffi.set_mode(mode)
    for nclip in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(nclip)]
        plans = [vc.vidcom2.CompressPlan(F, N, D, torch.bfloat16, dev, 0.25) for _ in range(nclip)]
        def step():
            for st, pl, x in zip(streams, plans, xs):
                with torch.cuda.stream(st):
                    pl.enqueue(x)
        for _ in range(5): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 40
        for _ in range(reps): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{mode}: {nclip} concurrent clips: {dt*1e6:.0f} us per round, {nclip*F*N/dt/1e6:.1f} M tokens/s")
