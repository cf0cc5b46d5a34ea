#!/usr/bin/env python
"""In-kernel wall-clock stamps of one pass (library built with -DVC2_DEBUG_TIMING): python scripts/dbg_timing.py [lib.so]"""
import ctypes, os, sys
if len(sys.argv) > 1:
    os.environ["VC2_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
F, N, D = 128, 196, 3584
x = synth.make(F, N, D, torch.bfloat16, 0, "drift").cuda()
_ffi.set_mode(os.environ.get('VC2_DBG_MODE', 'torch'))
plan = vc.vidcom2.CompressPlan(F, N, D, torch.bfloat16, x.device, 0.25)
L = ctypes.CDLL(_ffi.LIB_PATH)
t = (ctypes.c_ulonglong * 512)(); v = (ctypes.c_int * 512)(); n = ctypes.c_int(0)
for it in range(4):
    plan.enqueue(x); plan.finish()
    L.vc2_debug_read(t, v, ctypes.byref(n), 1)
rows = sorted((t[i], v[i]) for i in range(min(n.value, 512)))
t0 = rows[0][0]
print("stamps:", n.value)
for tt, vv in rows:
    print(f"{(tt - t0) / 100.0:8.2f} us  tag {vv}")
