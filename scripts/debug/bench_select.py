"""Micro-benchmark of the selection kernels (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi
from vidcom2_amd._ffi import lib, ptr, stream_ptr, check
dev = torch.device("cuda:0")

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for F, N in [(128, 196), (32, 196), (64, 324), (1, 196)]:
    for frac, name in [(1.0, "k=N"), (0.25, "k=N/4"), (0.01, "k small")]:
        for tie in (False, True):
            sc = torch.randn(F, N, device=dev)
            if tie: sc = (sc * 2).round() / 2
            scales = torch.full((F,), frac, device=dev)
            ws = torch.empty(F * N * 4 + F * 4 + 1024, dtype=torch.uint8, device=dev)
            ks = torch.empty(F, dtype=torch.int64, device=dev); offs = torch.empty(F + 1, dtype=torch.int64, device=dev)
            idx = torch.empty(F * N, dtype=torch.int64, device=dev); kout = torch.zeros(2, dtype=torch.int64, device=dev)
            fn = lambda: check(lib().vc2_select(ptr(sc), ptr(scales), F, N, 0, 2, 0, ptr(ws), ws.numel(), ptr(ks), ptr(offs), ptr(idx), F * N, ptr(kout), stream_ptr(dev)), "sel")
            print(f"select F={F} N={N} {name} ties={tie}: {timeit(fn):.1f} us (incl. 2 widen kernels)")
for D in (1024, 3584, 4096):
    for tie in (False, True):
        v = torch.rand(D, device=dev)
        if tie: v = (v * 40).round() / 40
        mask = torch.empty(D, dtype=torch.uint8, device=dev); cols = torch.empty(D, dtype=torch.int32, device=dev)
        fn = lambda: check(lib().vc2_chan_select(ptr(v), D, D // 2, ptr(mask), ptr(cols), None, None, None, None, stream_ptr(dev)), "cs")
        print(f"chan_select D={D} ties={tie}: {timeit(fn):.1f} us")
# empty kernel launch floor
x = torch.zeros(16, device=dev)
print("torch tiny kernel:", timeit(lambda: x.add_(1)), "us")
