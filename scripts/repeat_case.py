"""Run one case N times in the default mode and count runs that differ from the oracle (races show up as a rate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle as O
from vidcom2_amd import synth, _ffi
from vidcom2_amd.vidcom2 import compress
F, N, D, dn, dist, seed, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[dn]
x = synth.make(F, N, D, dt, seed, dist)
O.set_mode("torch")
o = O.compress_indices(x, N, 0.25)
xd = x.cuda()
bad = 0
for i in range(reps):
    r = compress(xd, N, 0.25, want_scores=True)
    if not (torch.equal(r.v_score.cpu(), o["v"]) and torch.equal(r.f_score.cpu(), o["f"])):
        bad += 1
print(os.environ.get("VC2_LIB_PATH", "default lib"), "runs", reps, "differing from the oracle:", bad)
