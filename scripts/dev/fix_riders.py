#!/usr/bin/env python
"""The norm-fix rider waves of the fused centre launch (k_frame_centres): begin / end per rider wave
(-DVC2_DEBUG_TIMING build): python scripts/dev/fix_riders.py lib.so [workload]"""
import ctypes, os, sys
os.environ["VC2_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg5clip"
F, N, D, dt = {"cfg5clip": (128, 196, 4096, torch.float16), "target": (128, 196, 3584, torch.bfloat16), "target_f16": (128, 196, 3584, torch.float16)}[wl]
x = synth.make(F, N, D, dt, 0, "drift").cuda()
plan = vc.vidcom2.CompressPlan(F, N, D, dt, x.device, 0.25)
L = ctypes.CDLL(_ffi.LIB_PATH)
buf = (ctypes.c_ulonglong * (8 * 2 * 4096))()
for it in range(6):
    plan.enqueue(x); plan.finish()
torch.cuda.synchronize()
L.vc2_debug_wg(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 2, 4096).astype(np.float64) / 100.0
b, e = a[5, 0, :4095], a[5, 1, :4095]
t0 = a[5, 0, 4095]
m = e > 0
pct = lambda v: " ".join(f"{np.percentile(v, q):6.1f}" for q in (0, 10, 50, 90, 99, 100))
print(f"{wl}: {m.sum()} rider waves had an entry; us from a frame workgroup's begin, p0/10/50/90/99/100")
print("  begin   :", pct((b - t0)[m]))
print("  end     :", pct((e - t0)[m]))
print("  duration:", pct((e - b)[m]))
# k_video_centre: main waves (slot 7) and replay rider waves (slot 6)
mb, me, rb, re_ = a[7, 0], a[7, 1], a[6, 0], a[6, 1]
mm, rm = me > 0, re_ > 0
if mm.any():
    t1 = min(mb[mm].min(), rb[rm].min() if rm.any() else 1e30)
    print(f"k_video_centre: {mm.sum()} main waves, {rm.sum()} rider waves; us from the first begin")
    print("  main  begin:", pct((mb - t1)[mm]), "| end:", pct((me - t1)[mm]))
    if rm.any():
        print("  rider begin:", pct((rb - t1)[rm]), "| end:", pct((re_ - t1)[rm]), "| duration:", pct((re_ - rb)[rm]))
tk = None
