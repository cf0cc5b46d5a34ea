import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oracle as O
import vidcom2_amd as vc
from vidcom2_amd import synth, _ffi
dev = torch.device("cuda:0")
O.set_mode("torch")
F, N, D, base = 16, 324, 3584, .125
x = synth.make(F, N, D, torch.bfloat16, 0, "drift"); xd = x.to(dev)
ref = O.compress_indices(x, N, base)
for mode in (0, 1, 2, 1, 1):
    _ffi.lib().vc2_set_mode(mode)
    got = vc.compress(xd, N, base, want_scores=True)
    bv = (got.v_score.cpu().double() != ref["v"].double()).nonzero().tolist(); bf = (got.f_score.cpu().double() != ref["f"].double()).nonzero().tolist()
    print("mode", mode, "v mism", bv[:5], "f mism", bf[:5], flush=True)
