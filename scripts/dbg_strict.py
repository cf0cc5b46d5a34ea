import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oracle as O
import vidcom2_amd as vc
from vidcom2_amd import synth, _ffi
dev = torch.device("cuda:0")
def cnt(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return int(((a != b) & ~(a.isnan() & b.isnan())).sum())
for (F, N, D, dn, seed, dist, base) in [(16, 324, 3584, "bf16", 0, "drift", .125), (16, 169, 3584, "f16", 0, "iid", .15), (32, 196, 3584, "f16", 2, "drift", .25)]:
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dn]
    x = synth.make(F, N, D, dt, seed, dist); xd = x.to(dev)
    for mode in ("exact", "torch"):
        O.set_mode(mode); _ffi.set_mode(mode)
        ref = O.compress_indices(x, N, base)
        got = vc.compress(xd, N, base, want_scores=True)
        bv = ((got.v_score.cpu().double() != ref["v"].double())).nonzero()
        bf = ((got.f_score.cpu().double() != ref["f"].double())).nonzero()
        print(dn, (F, N, D), mode, "v mism", len(bv), bv[:6].tolist(), "f mism", len(bf), bf[:6].tolist(), "idx equal", torch.equal(got.global_idx.cpu(), ref["global_idx"]), flush=True)
