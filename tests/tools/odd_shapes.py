"""GPU box: odd channel counts (C % 32 != 0 -> torch's row_sum path in the centre replay), all modes incl. the
always-replay debug mode, vs the oracle."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle as O
from vidcom2_amd import synth, _ffi
from vidcom2_amd.vidcom2 import compress
shapes = [(5, 36, 200), (7, 50, 72), (3, 17, 66), (16, 100, 1000), (9, 64, 24), (4, 300, 136), (40, 196, 328)]
bad = n = 0
for (F, N, D), dt, dist, mode in itertools.product(shapes, (torch.float16, torch.bfloat16, torch.float32), ("iid", "drift"),
                                                   (("torch", 1), ("torch", 2), ("exact", 0))):
    O.set_mode(mode[0]); _ffi.lib().vc2_set_mode(mode[1])
    for seed in range(3):
        x = synth.make(F, N, D, dt, seed, dist)
        r = compress(x.cuda(), N, 0.3, want_scores=True)
        o = O.compress_indices(x, N, 0.3)
        ok = torch.equal(r.global_idx.cpu(), o["global_idx"]) and torch.equal(r.ks.cpu(), o["ks"])
        if dt != torch.float32:
            ok = ok and torch.equal(r.v_score.cpu(), o["v"]) and torch.equal(r.f_score.cpu(), o["f"])
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", F, N, D, dt, dist, mode, seed)
_ffi.set_mode("torch")
print(n, "cases", bad, "mismatches")
