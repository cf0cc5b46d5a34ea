"""Where the host time of one vidcom2_compression() call goes (median of 30): plan construction, enqueue, finish."""
import statistics, time, torch
import vidcom2_amd as vc
from vidcom2_amd import synth
from vidcom2_amd.vidcom2 import CompressPlan
F, N, D = 128, 196, 3584
x = synth.make(F, N, D, torch.bfloat16, 0).cuda()
for _ in range(5):
    vc.vidcom2_compression(x, "llava_ov")
t = {"plan": [], "enqueue": [], "finish": [], "total": []}
for _ in range(30):
    torch.cuda.synchronize()
    a = time.perf_counter(); p = CompressPlan(F, N, D, x.dtype, x.device, 0.25)
    b = time.perf_counter(); p.enqueue(x)
    c = time.perf_counter(); r = p.finish()
    d = time.perf_counter()
    for k, v in zip(t, (b - a, c - b, d - c, d - a)):
        t[k].append(v * 1e6)
print({k: round(statistics.median(v), 1) for k, v in t.items()})
