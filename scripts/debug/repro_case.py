"""One (F, N, D, dtype, dist, seed) case: HIP modes against the oracle, where they differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle as O
from vidcom2_amd import synth, _ffi
from vidcom2_amd.vidcom2 import compress
F, N, D, dn, dist, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[dn]
x = synth.make(F, N, D, dt, seed, dist)
O.set_mode("torch")
o = O.compress_indices(x, N, 0.25)
O.set_mode("exact")
oe = O.compress_indices(x, N, 0.25)
print("oracle torch vs exact: v", int((o["v"] != oe["v"]).sum()), "f", int((o["f"] != oe["f"]).sum()))
for mode in ("torch", "torch_proven", 2, "exact"):
    if mode == 2:
        _ffi.lib().vc2_set_mode(2)
    else:
        _ffi.set_mode(mode)
    r = compress(x.cuda(), N, 0.25, want_scores=True)
    ref = oe if mode == "exact" else o
    dv = (r.v_score.cpu() != ref["v"]).nonzero()
    df = (r.f_score.cpu() != ref["f"]).nonzero()
    print("mode", mode, "v mismatches", dv.shape[0], dv[:4].tolist(), "f mismatches", df.shape[0], df[:4].tolist(),
          "idx equal", torch.equal(r.global_idx.cpu(), ref["global_idx"]))
    for t in dv[:3].tolist():
        print("   v", t, float(r.v_score.cpu()[tuple(t)]), float(ref["v"][tuple(t)]))
_ffi.set_mode("torch")
