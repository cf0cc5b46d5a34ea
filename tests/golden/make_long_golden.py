#!/usr/bin/env python
"""Long-clip goldens from the REFERENCE (build container only: needs /root/reference).

    python tests/golden/make_long_golden.py      ->  tests/golden/long_cases.json

Clips with more than 2^19 tokens: there torch's outer-sum cascade (ATen/native/cpu/SumKernel.cpp, level_power =
max(4, ceil_log2(n) / 4)) adds the video-centre mean in blocks of 32 rows instead of 16.  The `cancel` inputs make the
reference's own centre depend on that: the script records, per case, in how many video-centre columns the reference
differs from a cascade with level_power 4 and with level_power 5 (numpy restatement below, diagnostics only), so the
test knows which cases can tell the two apart.  Stored as digests (budgets in full); inputs are regenerated from the
seed at test time (vidcom2_amd/synth.py).  Nothing of the reference's source is copied.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from token_compressor.vidcom2 import vidcom2 as R  # noqa: E402  (the reference)

import oracle as O  # noqa: E402
from vidcom2_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
DT = {"bf16": torch.bfloat16, "f16": torch.float16}

CASES = [  # name, F, N, D, dtype, dist, seed
    ("long", 3000, 196, 256, "bf16", "cancel", 0),      # 588 000 tokens; the reference's centre is the lp = 5 cascade's
    ("long", 2680, 196, 256, "f16", "cancel", 1),       # 525 280: just past 2^19, a 5-block tail group
    ("long", 3000, 196, 256, "bf16", "drift", 0),
    ("long", 3000, 196, 256, "f16", "drift", 1),
    ("long_sharded", 3072, 196, 256, "bf16", "cancel", 13),  # 8 ranks x 384 frames: rows per rank % 32 == 0
    ("long_sharded", 3072, 196, 256, "f16", "cancel", 9),
    ("long_tailcols", 2700, 196, 200, "f16", "cancel", 0),   # C = 100: columns 96.. take row_sum's four chains
]


def cascade(xf: np.ndarray, lp: int) -> np.ndarray:
    """multi_row_sum over the rows of fp32 xf[n, C] with a given level_power (diagnostics)."""
    n, C = xf.shape
    B = 1 << lp
    nb = n // B
    blocks = xf[: nb * B].reshape(nb, B, C)
    bs = blocks[:, 0, :].copy()
    for u in range(1, B):
        bs += blocks[:, u, :]
    acc = [np.zeros(C, np.float32) for _ in range(4)]
    for b in range(nb):
        acc[0] = bs[b]
        i = (b + 1) * B
        for j in range(1, 4):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = np.zeros(C, np.float32)
            if i & ((B - 1) << (j * lp)):
                break
    for r in range(nb * B, n):
        acc[0] = acc[0] + xf[r]
    for j in range(1, 4):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def main():
    out = []
    for name, Fr, N, D, dn, dist, seed in CASES:
        x = synth.make(Fr, N, D, DT[dn], seed, dist)
        t0 = time.time()
        sel = R.select_low_var_channels(x)
        v, f = R.compute_gaussian_scores(sel, N)
        s = -v.mean(dim=-1)
        scales = R.compute_scales(s, 0.25)
        ks = (scales * N).round().long().clamp(min=1).tolist()
        idx = R.select_outlier_indices(v + f, scales, N)
        g = R._map_linear_offset(idx, N)
        dt_ref = time.time() - t0
        frames = F.normalize(sel.view(-1, N, sel.shape[-1]), dim=-1)
        vc_ref = frames.mean(dim=(0, 1))
        C = vc_ref.numel()
        full = (C // 32) * 32
        xf = frames.reshape(-1, C).float().numpy()
        n = xf.shape[0]
        vc_exact = (torch.from_numpy(xf.astype(np.float64).sum(0)).float() / float(n)).to(DT[dn])
        c4 = (torch.from_numpy(cascade(xf, 4)) / float(n)).to(DT[dn])
        c5 = (torch.from_numpy(cascade(xf, 5)) / float(n)).to(DT[dn])
        O.set_mode("torch")
        o = O.compress_indices(x, N, 0.25)
        rec = dict(name=name, F=Fr, N=N, D=D, base=0.25, dtype=dn, dist=dist, seed=seed,
                   x_sha256=synth.sha256_tensor(x), ks_sha256=synth.sha256_tensor(torch.tensor(ks)), ks_head=ks[:16],
                   K=int(g.numel()), idx_sha256=synth.sha256_tensor(g), idx_head=g[:8].tolist(), idx_tail=g[-8:].tolist(),
                   v_sha256=synth.sha256_tensor(v), f_sha256=synth.sha256_tensor(f),
                   video_centre=dict(ne_exact=int((vc_ref != vc_exact).sum()),
                                     ne_level_power_4=int((vc_ref[:full] != c4[:full]).sum()),
                                     ne_level_power_5=int((vc_ref[:full] != c5[:full]).sum())),
                   oracle_torch_mode=dict(ks=bool(o["ks"].tolist() == ks), idx=bool(torch.equal(o["global_idx"], g)),
                                          v_mismatch=int((o["v"].double() != v.double()).sum()),
                                          f_mismatch=int((o["f"].double() != f.double()).sum())))
        out.append(rec)
        print(name, Fr, N, D, dn, dist, seed, "K", rec["K"], f"reference {dt_ref:.1f}s", rec["video_centre"],
              rec["oracle_torch_mode"], flush=True)
        del x, sel, v, f, frames, xf
    with open(os.path.join(HERE, "long_cases.json"), "w") as fh:
        json.dump(dict(generator="tests/golden/make_long_golden.py", torch=torch.__version__,
                       threads=torch.get_num_threads(), cases=out), fh, indent=0)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
