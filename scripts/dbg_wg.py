#!/usr/bin/env python
"""Per-workgroup begin / end times of sweeps 2 and 3 (library built with -DVC2_DEBUG_TIMING):
python scripts/dbg_wg.py lib.so"""
import ctypes, os, sys
os.environ["VC2_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
F, N, D = 128, 196, 3584
x = synth.make(F, N, D, torch.bfloat16, 0, sys.argv[2] if len(sys.argv) > 2 else "drift").cuda()
plan = vc.vidcom2.CompressPlan(F, N, D, torch.bfloat16, x.device, 0.25)
L = ctypes.CDLL(_ffi.LIB_PATH)
buf = (ctypes.c_ulonglong * (8 * 2 * 4096))()
for it in range(5):
    plan.enqueue(x); plan.finish()
L.vc2_debug_wg(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 2, 4096).astype(np.float64) / 100.0
for slot, name in ((1, "k_norm_colsum"), (2, "k_dist")):
    b, e = a[slot, 0], a[slot, 1]
    m = e > 0
    t0 = b[m].min()
    b, e = b[m] - t0, e[m] - t0
    pct = lambda v: " ".join(f"{np.percentile(v, q):6.1f}" for q in (0, 10, 50, 90, 99, 100))
    print(f"{name}: {m.sum()} workgroups; begin p0/10/50/90/99/100: {pct(b)} | end: {pct(e)} | duration: {pct(e - b)}")
    late = np.argsort(e)[-8:]
    print("   last to end (wg: begin -> end):", ", ".join(f"{i}: {b[i]:.1f}->{e[i]:.1f}" for i in late))
# k_dist: row loop / phase 2 (replays) / phase 3 per workgroup
b, p2, p3, e = a[2, 0], a[0, 0], a[0, 1], a[2, 1]
m = e > 0
t0 = b[m].min()
rl, r2, r3 = (p2 - b)[m], (p3 - p2)[m], (e - p3)[m]
pct = lambda v: " ".join(f"{np.percentile(v, q):6.1f}" for q in (0, 10, 50, 90, 99, 100))
print("k_dist row loop:", pct(rl), "| phase 2:", pct(r2), "| phase 3:", pct(r3))
busy = r2 > 1.0
print(f"   workgroups with replays: {busy.sum()}; their phase 2: {pct(r2[busy]) if busy.any() else ''}; row-loop end of those: {pct((p2 - t0)[m][busy]) if busy.any() else ''}")
sqd, smd = a[3, 0], a[3, 1]
one = busy & ((sqd - p2)[m] > 0) & ((sqd - p2)[m] < 50)
if one.any():
    print("   first entry: squares ready after", pct((sqd - p2)[m][one]), "| cascade sum + barrier", pct((smd - sqd)[m][one]))
# structure of the row-loop time: by XCD (blockIdx % 8), by split (blockIdx % 8 is also the split when S2 = 8!), by frame
idx = np.arange(4096)[m]
for name, key in (("blockIdx % 8", idx % 8), ("(blockIdx // 8) % 16", (idx // 8) % 16), ("blockIdx // 256", idx // 256)):
    print("   row loop by", name, ":", " ".join(f"{rl[key == k].mean():.1f}" for k in np.unique(key)))
