"""Run HERE (needs /root/reference; never on the GPU box): does the reference's output depend on torch's thread
count?  The fixtures were generated with 8 threads; this reruns a few shapes with 1, 3 and 8 threads and compares
scores, budgets and kept indices bit for bit."""
import hashlib
import sys

import torch

sys.path.insert(0, "/root/reference")
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from token_compressor.vidcom2 import vidcom2 as R          # noqa: E402
from vidcom2_amd import synth                               # noqa: E402


def run(x, tpf, base):
    sel = R.select_low_var_channels(x)
    v, f = R.compute_gaussian_scores(sel, tpf)
    scales = R.compute_scales(-v.mean(dim=-1), base)
    idx = R.select_outlier_indices(v + f, scales, tpf)
    h = hashlib.sha256()
    for t in (v, f, scales, torch.cat(idx)):
        h.update(t.contiguous().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()


cases = [(32, 196, 3584, torch.bfloat16, 0.25, "drift"), (32, 196, 3584, torch.float16, 0.25, "iid"),
         (128, 196, 3584, torch.bfloat16, 0.25, "drift"), (64, 324, 3584, torch.bfloat16, 0.125, "iid"),
         (16, 169, 1152, torch.float16, 0.25, "drift")]
bad = 0
for F, N, D, dt, base, dist in cases:
    x = synth.make(F, N, D, dt, 1, dist)
    digests = {}
    for nt in (1, 3, 8):
        torch.set_num_threads(nt)
        digests[nt] = run(x, N, base)
    same = len(set(digests.values())) == 1
    bad += not same
    print(F, N, D, dt, dist, "identical for 1 / 3 / 8 threads" if same else f"DIFFERS: {digests}")
print("thread-count dependent" if bad else "thread-count independent on all cases")
