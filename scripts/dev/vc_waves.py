#!/usr/bin/env python
"""k_video_centre wave by wave (library built with -DVC2_DEBUG_TIMING): main waves -- begin, flags known, level-1 groups stored
(waves of a block with a flagged column only), end; rider waves -- entries taken, duration.
    python scripts/dev/vc_waves.py lib_dbg.so [target|target_f16|cfg5clip]"""
import ctypes, os, sys
os.environ["VC2_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
wl = sys.argv[2] if len(sys.argv) > 2 else "target_f16"
F, N, D, dt = {"cfg5clip": (128, 196, 4096, torch.float16), "target": (128, 196, 3584, torch.bfloat16),
               "target_f16": (128, 196, 3584, torch.float16)}[wl]
x = synth.make(F, N, D, dt, 0, "drift").cuda()
plan = vc.vidcom2.CompressPlan(F, N, D, dt, x.device, 0.25)
L = ctypes.CDLL(_ffi.LIB_PATH)
for it in range(6):
    plan.enqueue(x); plan.finish()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 2 * 4096))()
L.vc2_debug_wg(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 2, 4096).astype(np.float64) / 100.0
vbuf = (ctypes.c_ulonglong * (6 * 4096))()
assert L.vc2_debug_vc(vbuf) == 0
v = np.frombuffer(vbuf, dtype=np.uint64).reshape(6, 4096)
mb, me, rb, re_ = a[7, 0], a[7, 1], a[6, 0], a[6, 1]
mm, rm = me > 0, re_ > 0
pct = lambda q: " ".join(f"{np.percentile(q, p):6.1f}" for p in (0, 10, 50, 90, 99, 100)) if len(q) else "-"
t1 = mb[mm].min()
fl = v[0].astype(np.float64) / 100.0
st = v[1].astype(np.float64) / 100.0
nfl = (v[2] & np.uint64(0xFFFFFFFF)).astype(int)
nvc = (v[2] >> np.uint64(32)).astype(int)
print(f"{wl}: {mm.sum()} main waves ({int(nvc[mm].max())} corrected norms in the pass), {rm.sum()} rider waves; p0/10/50/90/99/100 us")
print("  main  begin -> flags known:", pct((fl - mb)[mm]), "| end (from the first begin):", pct((me - t1)[mm]))
rep = mm & (nfl > 0)
print(f"  waves of blocks with flagged columns: {rep.sum()}; flagged columns per block: {dict(zip(*np.unique(nfl[rep], return_counts=True)))}")
for k in sorted(set(nfl[rep].tolist())):
    s_ = rep & (nfl == k) & (st > 0)
    print(f"    {k} column(s): flags -> groups stored", pct((st - fl)[s_]), "| groups stored -> end", pct((me - st)[s_]))
ent = v[3].astype(int)
c1 = int((v[4] & np.uint64(0xFFFFFFFF))[rm].max()); c2 = int((v[4] >> np.uint64(32))[rm].max())
print(f"  riders: {c1} listed means + {c2} correction entries; entries per rider wave {dict(zip(*np.unique(ent[rm], return_counts=True)))}")
for k in sorted(set(ent[rm].tolist())):
    s_ = rm & (ent == k)
    print(f"    {k} entries: duration", pct((re_ - rb)[s_]))
