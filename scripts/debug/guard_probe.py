import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidcom2_amd as vc
from vidcom2_amd import synth
from vidcom2_amd._ffi import lib
dev = torch.device("cuda:0")
out = (ctypes.c_int32 * 8)()
for (F, N, D, dt) in [(1, 196, 512, torch.bfloat16), (6, 196, 256, torch.bfloat16), (8, 196, 1024, torch.bfloat16), (32, 196, 3584, torch.bfloat16)]:
    x = synth.make(F, N, D, dt, 0, "iid").to(dev)
    r = vc.compress(x, N, 0.25)
    lib().vc2_selftest_counters(out, 1)
    print(F, N, D, "K", r.K, "guard hits [coop-select, solo-select, sort-level, sort-queue, sort-subtree]:", list(out)[:5], flush=True)
