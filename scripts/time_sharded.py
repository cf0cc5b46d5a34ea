import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidcom2_amd import synth, _ffi
from vidcom2_amd.sharded import ShardedCompressor
import vidcom2_amd as vc
F, N, D = 128, 196, 3584
x = synth.make(F, N, D, torch.bfloat16, 0).cuda()
for side in (1, 0):
    _ffi.lib().vc2_set_side_stream(side)
    sc = ShardedCompressor(F, N, D, torch.bfloat16, x.device, 0.25, group=None)
    for _ in range(10): sc.enqueue(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): sc.enqueue(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"sharded side={side}: cpu enqueue {(t1-t0)/50*1e6:.0f} us/pass, total {(t2-t0)/50*1e6:.0f} us/pass")
_ffi.lib().vc2_set_side_stream(1)
plan = vc.vidcom2.CompressPlan(F, N, D, torch.bfloat16, x.device, 0.25)
for _ in range(10): plan.enqueue(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): plan.enqueue(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"fused: cpu enqueue {(t1-t0)/50*1e6:.0f} us/pass, total {(t2-t0)/50*1e6:.0f} us/pass")
