"""Container-only soak: reference (imported from /root/reference) vs oracle (torch mode) on fresh seeds.
Prints every case whose kept indices / budgets / half-precision scores differ.  Not a test (needs the reference)."""
import sys, time, itertools
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, "/root/reference")
import torch
from token_compressor.vidcom2 import vidcom2 as R
import oracle as O
from vidcom2_amd import synth
torch.set_grad_enabled(False)
O.set_mode("torch")
shapes = [(32, 196, 3584), (16, 169, 1152), (24, 144, 2048), (8, 324, 3584), (64, 196, 896), (12, 100, 1280)]
seeds = range(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 106)
bad = 0; n = 0
t0 = time.time()
for (F, N, D), dt, dist, seed in itertools.product(shapes, (torch.float16, torch.bfloat16), ("iid", "drift"), seeds):
    x = synth.make(F, N, D, dt, seed, dist)
    sel = R.select_low_var_channels(x)
    v, f = R.compute_gaussian_scores(sel, N)
    scales = R.compute_scales(-v.mean(dim=-1), 0.25)
    idx = R._map_linear_offset(R.select_outlier_indices(v + f, scales, N), N)
    o = O.compress_indices(x, N, 0.25)
    ok = torch.equal(o["global_idx"], idx) and torch.equal(o["v"], v) and torch.equal(o["f"], f)
    n += 1
    if not ok:
        bad += 1
        print("MISMATCH", F, N, D, dt, dist, seed, "idx", torch.equal(o["global_idx"], idx), "v", int((o["v"] != v).sum()),
              "f", int((o["f"] != f).sum()), flush=True)
print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f}s")
