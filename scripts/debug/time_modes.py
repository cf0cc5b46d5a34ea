"""Per-pass time of the target shape in a given mode (A/B builds: VC2_LIB_PATH)."""
import sys, time, torch
sys.path.insert(0, ".")
import vidcom2_amd as vc
from vidcom2_amd.vidcom2 import CompressPlan
from vidcom2_amd import _ffi, synth
for dist in ("drift", "iid"):
    x = synth.make(128, 196, 3584, torch.bfloat16, 0, dist).cuda()
    for mode in sys.argv[1:] or ["torch", "torch_proven"]:
        _ffi.set_mode(mode)
        plan = CompressPlan(128, 196, 3584, torch.bfloat16, x.device, 0.25)
        for _ in range(300): plan.enqueue(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): plan.enqueue(x)
        torch.cuda.synchronize()
        print(dist, mode, "%.1f us" % ((time.perf_counter() - t0) / 200 * 1e6), flush=True)
