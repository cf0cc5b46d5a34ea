"""HIP path (torch-order mode) against the golden vectors captured from the reference itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import vidcom2_amd as vc
from vidcom2_amd import synth, _ffi
from conftest import load_core_cases, make_input
dev = torch.device("cuda:0")
_ffi.set_mode("torch")
ok = tot = 0
for c in load_core_cases():
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"]).to(dev)
    g = vc.compress(x, c["N"], c["base"], want_scores=True)
    ks = g.ks.cpu().tolist() == c["ks"]
    idx = g.global_idx.cpu().tolist() == c["global_idx"]
    if c["dtype"] == "f32":
        sv = sf = True
    else:
        sv = synth.sha256_tensor(g.v_score) == c["v_sha256"]; sf = synth.sha256_tensor(g.f_score) == c["f_sha256"]
    good = ks and idx and sv and sf
    tot += 1; ok += good
    if not good:
        print("MISMATCH", c["name"], c["dtype"], c["dist"], c["seed"], dict(ks=ks, idx=idx, v=sv, f=sf), flush=True)
print(f"{ok}/{tot} fixtures: budgets + kept indices (+ half-precision score digests) equal to the reference")
