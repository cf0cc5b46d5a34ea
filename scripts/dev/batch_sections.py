#!/usr/bin/env python
"""compress_batch's sections on the host clock (same-shape batch): setup / enqueue loop / joins + new outputs / the one sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidcom2_amd import synth, vidcom2 as V
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else torch.bfloat16
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3584
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
F, N = 128, 196
clips = [synth.make(F, N, D, dt, sd, "drift").cuda() for sd in range(4)]
batch = [clips[i % 4] for i in range(24)]
dev = clips[0].device
def once(report):
    t = [time.perf_counter()]
    cur = torch.cuda.current_stream(dev)
    lanes = [cur] + V._lane_streams(dev, k - 1)
    plans = []
    for st in lanes:
        with torch.cuda.stream(st):
            plans.append(V._cached_plan(F, N, D, dt, dev, 0.25, "linear", 0, False, True, 0))
    p0 = plans[0]; n = len(batch)
    idx_all = torch.empty((n, p0.cap), dtype=torch.int64, device=dev)
    ks_all = torch.empty((n, F), dtype=torch.int64, device=dev)
    rows_all = torch.empty((n, p0.cap, D), dtype=dt, device=dev)
    kout_all = torch.empty((n, 2), dtype=torch.int64, device=dev)
    for st in lanes[1:]:
        st.wait_stream(cur)
        for tt in (idx_all, ks_all, rows_all, kout_all): tt.record_stream(st)
    t.append(time.perf_counter())
    for i, x in enumerate(batch):
        plan = plans[i % k]
        plan.idx, plan.ks, plan.rows, plan.v, plan.f, plan.kout = idx_all[i], ks_all[i], rows_all[i], None, None, kout_all[i]
        plan.enqueue(x, stream=lanes[i % k])
    t.append(time.perf_counter())
    for st in lanes[1:]: cur.wait_stream(st)
    for plan in plans: plan.new_outputs()
    t.append(time.perf_counter())
    words = kout_all.tolist()
    t.append(time.perf_counter())
    if report:
        d = [(b - a) * 1e6 for a, b in zip(t, t[1:])]
        print(f"k={k} cap={p0.cap}: setup {d[0]:.0f} us | enqueue loop {d[1]:.0f} us ({d[1] / n:.1f} per clip) | joins + new outputs {d[2]:.0f} us | sync {d[3]:.0f} us | total {sum(d):.0f} us = {sum(d) / n:.1f} per clip")
for i in range(6): once(i >= 3)
