#!/usr/bin/env python
"""Sweep 2 (k_norm_colsum2) per workgroup: launch -> first row landed -> row loops over -> end (library built with
-DVC2_DEBUG_TIMING):  python scripts/dev/s2_wg.py lib.so [F]"""
import ctypes, os, sys
os.environ["VC2_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
F, N, D = int(sys.argv[2]) if len(sys.argv) > 2 else 128, 196, 3584
x = synth.make(F, N, D, torch.bfloat16, 0, "drift").cuda()
plan = vc.vidcom2.CompressPlan(F, N, D, torch.bfloat16, x.device, 0.25)
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
pct = lambda v: " ".join(f"{np.percentile(v, q):6.1f}" for q in (0, 10, 50, 90, 99, 100))
stream = m & (first > 0)
print(f"{m.sum()} workgroups ({stream.sum()} streaming, {m.sum() - stream.sum()} riders); us from the first begin, p0/10/50/90/99/100")
print("  begin           :", pct((b - t0)[m]))
print("  riders end      :", pct((e - t0)[m & ~stream]) if (m & ~stream).any() else "-")
print("  first row landed:", pct((first - t0)[stream]))
print("  row loops over  :", pct((loop - t0)[stream]))
print("  end             :", pct((e - t0)[stream]))
print("  loop duration   :", pct((loop - first)[stream]), "| tail (combine + stores):", pct((e - loop)[stream]))
idx = np.arange(4096)
for name, key in (("blockIdx % 8 (XCD)", idx % 8),):
    print("  loop end by", name, ":", " ".join(f"{(loop - t0)[stream & (key == k)].mean():.1f}" for k in range(8)))
rid = np.where(m & ~stream)[0]
if len(rid):
    print("  rider end times by rider index:", " ".join(f"{(e - t0)[i]:.1f}" for i in rid))
# streaming workgroups: does a frame boundary inside the chunk cost?  which chunks are slow?
nr = int((m & ~stream).sum())
q = -(-F * N // int(stream.sum()))
ids = np.where(stream)[0]
chunk = ids - nr
has_b = np.array([(c * q) // N != (min(F * N, (c + 1) * q) - 1) // N for c in chunk])
dur = (loop - first)[ids]
print(f"  chunks of {q} rows; with a frame boundary inside: {has_b.sum()}, loop duration {dur[has_b].mean():.1f} us; without: {dur[~has_b].mean():.1f} us")
slow = ids[np.argsort(dur)[-12:]]
print("  slowest loops (blockIdx: duration, begin->first row):", ", ".join(f"{i}: {(loop - first)[i]:.1f} ({(first - b)[i]:.1f})" for i in slow))
h = np.histogram(dur, bins=np.arange(12, 28, 1.0))
print("  loop duration histogram (1 us bins from 12):", " ".join(str(v) for v in h[0]))
