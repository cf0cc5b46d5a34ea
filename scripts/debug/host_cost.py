"""Host-side cost per clip of compress_batch's loop pieces."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidcom2_amd import synth, vidcom2 as V
F, N, D, dt = 128, 196, 3584, torch.bfloat16
x = synth.make(F, N, D, dt, 0, "drift").cuda()
plan = V._cached_plan(F, N, D, dt, x.device, 0.25, "linear", 0, False, True, 0)
def T(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize(); return t
print("cached_plan (incl. new_outputs): %.1f us" % T(lambda: V._cached_plan(F, N, D, dt, x.device, 0.25, "linear", 0, False, True, 0)))
print("new_outputs alone: %.1f us" % T(plan.new_outputs))
print("enqueue (host time, async, 4 calls): %.1f us" % T(lambda: plan.enqueue(x), 4)); print("enqueue (host time, async, 16 calls): %.1f us" % T(lambda: plan.enqueue(x), 16))
print("take: %.1f us" % T(plan.take))
