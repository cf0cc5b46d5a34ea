#!/usr/bin/env python
"""Fixtures for the standalone `_multi_scale_gaussian` helper (vidcom2.py:59-62), made by calling the
REFERENCE's function (container only):  python tests/golden/make_msg_golden.py  ->  msg_cases.json

Inputs are synthetic (`msg_inputs` in tests/_stub_models.py regenerates them from the seed); stored are
the sha256 of the reference's output and a few sampled values (fp32 is compared within 1e-5)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

from token_compressor.vidcom2 import vidcom2 as R  # noqa: E402

import _stub_models as S  # noqa: E402
from vidcom2_amd import synth  # noqa: E402

PAPER = [2 ** k for k in range(-3, 2)]
CASES = []
for dt in ("f32", "bf16", "f16"):
    for (F, N, C, seed) in [(8, 196, 1792, 1), (5, 36, 100, 2), (3, 17, 33, 3), (16, 169, 896, 4), (2, 7, 8, 5)]:
        for kind in ("video", "frame"):
            CASES.append(dict(F=F, N=N, C=C, dt=dt, seed=seed, kind=kind, alphas=PAPER))
    CASES.append(dict(F=4, N=50, C=256, dt=dt, seed=6, kind="frame", alphas=[0.3, 1, 7]))
    CASES.append(dict(F=4, N=50, C=256, dt=dt, seed=6, kind="video", alphas=[0.5]))

out = []
for c in CASES:
    x, cen = S.msg_inputs(c)
    ref = R._multi_scale_gaussian(x, cen, c["alphas"])
    flat = ref.reshape(-1)
    pick = torch.linspace(0, flat.numel() - 1, 16).long()
    out.append(dict(c, sha=synth.sha256_tensor(ref), sample_idx=pick.tolist(),
                    sample=[float(v) for v in flat[pick].double()], nonzero=int((flat != 0).sum())))
    print(c["dt"], c["F"], c["N"], c["C"], c["kind"], "nonzero", out[-1]["nonzero"], "/", flat.numel())
json.dump(out, open(os.path.join(HERE, "msg_cases.json"), "w"))
print("wrote msg_cases.json", len(out))
