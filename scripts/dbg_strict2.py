import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ctypes
import oracle as O
import vidcom2_amd as vc
from vidcom2_amd import synth, _ffi
dev = torch.device("cuda:0")
O.set_mode("torch"); _ffi.lib().vc2_set_mode(2)
for (F, N, D, dn, seed, dist, base) in [(4, 49, 64, "bf16", 0, "drift", .25), (8, 196, 1024, "bf16", 0, "drift", .25), (4, 100, 3584, "bf16", 0, "drift", .25), (4, 100, 3584, "f16", 0, "drift", .25), (4, 50, 4096, "bf16", 1, "iid", .25), (3, 40, 200, "bf16", 1, "iid", .25)]:
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dn]
    x = synth.make(F, N, D, dt, seed, dist); xd = x.to(dev)
    ref = O.compress_indices(x, N, base)
    got = vc.compress(xd, N, base, want_scores=True)
    bv = int((got.v_score.cpu().double() != ref["v"].double()).sum()); bf = int((got.f_score.cpu().double() != ref["f"].double()).sum())
    print(dn, (F, N, D), "always-replay: v mism", bv, "f mism", bf, "of", F * N, flush=True)
