"""Where the host time of one vidcom2_compression() call goes (median of 40), and the one-shot latency with / without
the spare outputs and the polled finish: python scripts/one_shot_breakdown.py"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vidcom2_amd as vc
from vidcom2_amd import synth, vidcom2 as V
F, N, D = 128, 196, 3584
x = synth.make(F, N, D, torch.bfloat16, 0).cuda()


def one_shot(n=40):
    lat = []
    for _ in range(n + 5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = vc.vidcom2_compression(x, model="qwen2_5_vl", base_scale=0.25, frame_token_len=N)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e6)
        del r
    return round(statistics.median(lat[5:]), 1)


for pre, early in ((False, False), (True, False), (True, True)):
    V._PREALLOC, V._EARLY_COUNT = pre, early
    V.clear_plan_cache()
    print(f"spare outputs {pre!s:5} early count {early!s:5}: one-shot {one_shot()} us (incl. the final device sync)")
plan = V._cached_plan(F, N, D, x.dtype, x.device, 0.25, "linear", 0, False, True, 0)
t = {"cached_plan": [], "enqueue(host)": [], "spare": [], "finish": [], "total": []}
for _ in range(40):
    torch.cuda.synchronize()
    a = time.perf_counter(); p = V._cached_plan(F, N, D, x.dtype, x.device, 0.25, "linear", 0, False, True, 0)
    b = time.perf_counter(); p.enqueue(x, mirror=True)
    c = time.perf_counter(); p.prepare_spare()
    d = time.perf_counter(); r = p.finish()
    e = time.perf_counter()
    for k, v in zip(t, (b - a, c - b, d - c, e - d, e - a)):
        t[k].append(v * 1e6)
print({k: round(statistics.median(v), 1) for k, v in t.items()})
