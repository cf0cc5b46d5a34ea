#!/usr/bin/env python
"""Sweep 2, general kernel against the streamlined one, array by array (library built with -DVC2_DEBUG_EXPORTS):
python scripts/dev/s2_compare.py lib.so [F N D dtype dist]"""
import ctypes, os, sys
os.environ["VC2_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
a = sys.argv[2:]
F, N, D = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (8, 196, 1024)
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a[3] if len(a) > 3 else "bf16"]
dist = a[4] if len(a) > 4 else "drift"
x = synth.make(F, N, D, dt, 0, dist).cuda()
L = ctypes.CDLL(_ffi.LIB_PATH)
off = (ctypes.c_int64 * 16)()
L.vc2_debug_plan.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
assert L.vc2_debug_plan(F, N, D, {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dt], off) == 0
o_den, o_part, o_rflag, o_tk, o_fixq, S, S_q, S_W, o_fc, o_vc, o_total = [int(v) for v in off[:11]]
C = D // 2
res = {}
for v2 in (0, 1):
    L.vc2_debug_set(0, v2); L.vc2_debug_set(1, v2)
    plan = vc.vidcom2.CompressPlan(F, N, D, dt, x.device, 0.25)
    plan.enqueue(x); plan.finish(); torch.cuda.synchronize()
    ws = plan.ws.view(torch.uint8)
    g = lambda o, n, t: ws[o:o + n].clone().view(t).cpu()
    res[v2] = dict(den=g(o_den, F * N * 4, torch.float32), part=g(o_part, F * S * C * 8, torch.float64),
                   rflag=g(o_rflag, F * N, torch.uint8), tk=g(o_tk, 64 * 4, torch.int32), fc=g(o_fc, F * C * 4, torch.float32),
                   vc=g(o_vc, C * 4, torch.float32), total=g(o_total, F * N * 4, torch.float32))
print(f"F={F} N={N} D={D} {dt} {dist}: S={S} S_q={S_q} S_W={S_W}")
for k in ("den", "part", "rflag", "fc", "vc", "total"):
    a0, a1 = res[0][k], res[1][k]
    ne = ~((a0 == a1) | ((a0 != a0) & (a1 != a1)))
    print(f"  {k:6s}: {int(ne.sum())} of {a0.numel()} differ", (ne.nonzero().flatten()[:8].tolist(), a0[ne][:4].tolist(), a1[ne][:4].tolist()) if ne.any() else "")
print("  tickets general:", res[0]["tk"][:8].tolist(), "streamlined:", res[1]["tk"][:8].tolist())
