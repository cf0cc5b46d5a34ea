"""Round-2 probe: queue counters of one pass at several shapes (how many tokens are flagged / corrected)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidcom2_amd as vc
from vidcom2_amd import synth, _ffi
from vidcom2_amd._ffi import lib, ptr, DTYPE_CODE
dev = torch.device("cuda:0")
for (F, N, D, dt, base, dist) in [(128, 196, 3584, torch.bfloat16, .25, "drift"), (128, 196, 3584, torch.bfloat16, .25, "iid"),
                                  (32, 196, 3584, torch.bfloat16, .25, "drift"), (64, 324, 3584, torch.bfloat16, .125, "drift"),
                                  (128, 196, 4096, torch.float16, .25, "drift"), (32, 196, 3584, torch.float16, .25, "iid")]:
    for seed in (0, 1):
        x = synth.make(F, N, D, dt, seed, dist).to(dev)
        plan = vc.vidcom2.CompressPlan(F, N, D, dt, dev, base)
        plan.enqueue(x); r = plan.finish()
        out = (ctypes.c_int32 * 8)()
        rc = lib().vc2_pass_counters(F, N, D, DTYPE_CODE[dt], ptr(plan.ws), out)
        print(f"{F}x{N}x{D} {dt} {dist} s{seed}: K={r.K} rc={rc} [spare, dist_fix, norm_fix, norm_corr, cfix, vfix, -, -] = {list(out)}", flush=True)
