import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidcom2_amd import synth, _ffi
import vidcom2_amd as vc
F, N, D = 128, 196, 3584
x = synth.make(F, N, D, torch.bfloat16, 0).cuda()
plan = vc.vidcom2.CompressPlan(F, N, D, torch.bfloat16, x.device, 0.25)
for _ in range(5): plan.enqueue(x)
ref = plan.finish()
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("eager: %.1f us/pass" % timeit(lambda: plan.enqueue(x)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): plan.enqueue(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        plan.enqueue(x)
    print("captured")
    print("graph: %.1f us/pass" % timeit(lambda: g.replay()))
    g.replay(); r = plan.finish()
    print("same result:", torch.equal(r.global_idx, ref.global_idx), torch.equal(r.rows, ref.rows))
except Exception as e:
    print("capture failed:", repr(e)[:300])
