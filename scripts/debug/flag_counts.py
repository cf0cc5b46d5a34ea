#!/usr/bin/env python
"""How many centre means / norms / distances a pass replays: python scripts/flag_counts.py [dist] [dtype]"""
import ctypes, sys
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
dist = sys.argv[1] if len(sys.argv) > 1 else "drift"
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
for (F, N, D) in ((128, 196, 3584), (32, 196, 3584), (64, 324, 3584)):
    x = synth.make(F, N, D, dt, 0, dist).cuda()
    plan = vc.vidcom2.CompressPlan(F, N, D, dt, x.device, 0.25)
    plan.enqueue(x); plan.finish()
    out = (ctypes.c_int32 * 8)()
    _ffi.check(_ffi.lib().vc2_pass_counters(F, N, D, _ffi.DTYPE_CODE[dt], _ffi.ptr(plan.ws), out), "counters")
    print(f"{F}x{N}x{D} {dist}: norm rows queued {out[2]} (corrected {out[3]}), video-centre columns replayed {out[5]} of {D // 2}, "
          f"frame means replayed {out[7]} of {F * (D // 2)}")
