"""GPU-box soak: HIP path (flag-and-fix replay) vs the oracle (replays everything) on fresh seeds, torch mode."""
import sys, time, itertools, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle as O
from vidcom2_amd import synth, _ffi
from vidcom2_amd.vidcom2 import compress
O.set_mode("torch"); _ffi.set_mode("torch")
shapes = [(32, 196, 3584), (16, 169, 1152), (24, 144, 2048), (8, 324, 3584), (64, 196, 896), (12, 100, 1280),
          (128, 196, 3584)]
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = n = 0
t0 = time.time()
for (F, N, D), dt, dist, seed in itertools.product(shapes, (torch.float16, torch.bfloat16), ("iid", "drift"), range(lo, hi)):
    if F == 128 and seed >= lo + 3:
        continue
    x = synth.make(F, N, D, dt, seed, dist)
    r = compress(x.cuda(), N, 0.25, want_scores=True)
    o = O.compress_indices(x, N, 0.25)
    ok = (torch.equal(r.global_idx.cpu(), o["global_idx"]) and torch.equal(r.v_score.cpu(), o["v"])
          and torch.equal(r.f_score.cpu(), o["f"]) and torch.equal(r.ks.cpu(), o["ks"]))
    n += 1
    if n % 1000 == 0:
        print(f"{n} cases, {bad} mismatches so far, {time.time() - t0:.0f}s", flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", F, N, D, dt, dist, seed, int((r.v_score.cpu() != o["v"]).sum()), int((r.f_score.cpu() != o["f"]).sum()), flush=True)
print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f}s")
