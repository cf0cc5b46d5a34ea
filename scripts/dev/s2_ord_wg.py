#!/usr/bin/env python
"""Sweep 2, ORD geometry (OrdGeo: unequal pieces): per-workgroup row-loop end times by class -- riders, the larger
workgroups (listed first), the smaller ones -- to check the pairing the launch order assumes (workgroup i and i + 256 on one CU).
Library built with -DVC2_DEBUG_TIMING:  python scripts/dev/s2_ord_wg.py lib.so [F N]"""
import ctypes, os, sys
os.environ["VC2_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 196
D = 3584
x = synth.make(F, N, D, torch.bfloat16, 0, "drift").cuda()
plan = vc.vidcom2.CompressPlan(F, N, D, torch.bfloat16, x.device, 0.125 if N == 324 else 0.25)
L = ctypes.CDLL(_ffi.LIB_PATH)
buf = (ctypes.c_ulonglong * (8 * 2 * 4096))()
for it in range(8):
    plan.enqueue(x); plan.finish()
torch.cuda.synchronize()
L.vc2_debug_wg(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 2, 4096).astype(np.float64) / 100.0
b, e, first, loop = a[1, 0], a[1, 1], a[4, 0], a[4, 1]
m = e > 0
t0 = b[m].min()
stream = m & (first > 0)
nr = int((m & ~stream).sum())
ids = np.where(stream)[0]
pct = lambda v: " ".join(f"{np.percentile(v, q):6.1f}" for q in (0, 10, 50, 90, 100)) if len(v) else "-"
print(f"{m.sum()} workgroups: {nr} riders, {len(ids)} streaming; kernel end {(e - t0)[m].max():.1f} us; p0/10/50/90/100")
print("  riders end       :", pct((e - t0)[m & ~stream]))
dur = (loop - first)
# classes by loop duration jump: the larger workgroups come first in the launch
d = dur[ids]
big = ids[d > (d.min() + d.max()) / 2] if d.max() > 1.15 * d.min() else ids[:0]
print("  all   loop end   :", pct((loop - t0)[ids]), "| duration:", pct(d))
for lo_, hi_, name in ((nr, 256, "blockIdx < 256 (listed first)"), (256, 256 + nr, f"256 .. {256 + nr - 1} (CU-mates of riders?)"),
                       (256 + nr, 4096, "the rest")):
    sel = ids[(ids >= lo_) & (ids < hi_)]
    if len(sel):
        print(f"  {name:38s}: n={len(sel):3d} loop end", pct((loop - t0)[sel]), "| duration", pct(dur[sel]), "| end", pct((e - t0)[sel]))
# does i pair with i + 256?  compare the summed loop durations of (i, i + 256)
pairs = [(i, i + 256) for i in ids if i + 256 in set(ids.tolist())]
if pairs:
    s = np.array([dur[i] + dur[j] for i, j in pairs])
    print(f"  {len(pairs)} pairs (i, i + 256): sum of the two loop durations", pct(s), "| later end of the two", pct(np.array([max((loop - t0)[i], (loop - t0)[j]) for i, j in pairs])))
print("  loop end by XCD (blockIdx % 8):", " ".join(f"{(loop - t0)[stream & (np.arange(4096) % 8 == k)].mean():.1f}" for k in range(8)))
