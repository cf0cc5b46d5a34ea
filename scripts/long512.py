"""Timing of the 512-frame clip (720 MB) for a library build: VC2_LIB_PATH=... python scripts/long512.py
(four 128-frame `drift` clips in a row, three different ones -- what bench.py's long512 leg uses)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidcom2_amd as vc
from vidcom2_amd import synth
F, N, D = 512, 196, 3584
xs = [synth.make(128, N, D, torch.bfloat16, seed=sd, dist="drift").cuda() for sd in (0, 1, 2)]
x = torch.cat([xs[0], xs[1], xs[2], xs[0]])
del xs
p = vc.vidcom2.CompressPlan(F, N, D, torch.bfloat16, x.device, 0.25)
for _ in range(5):
    p.enqueue(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    p.enqueue(x)
torch.cuda.synchronize()
print("512 frames: %.1f us / pass" % ((time.perf_counter() - t0) / 20 * 1e6))
