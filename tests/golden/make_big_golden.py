#!/usr/bin/env python
"""Large-shape goldens from the REFERENCE (build container only: needs /root/reference).

    python tests/golden/make_big_golden.py

Writes tests/golden/big_cases.json: the reference's result at BASELINE.json's big configurations -- cfg4
(512 x 196 x 3584 bf16, the frame-sharded one), and extra seeds / distributions at the headline target shape and
cfg3 -- stored as digests (budgets in full; kept indices, kept rows and both score tensors as sha256).  Inputs are
regenerated from the seed at test time (vidcom2_amd/synth.py); nothing of the reference's source is copied.
"""
from __future__ import annotations

import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from token_compressor.vidcom2 import vidcom2 as R  # noqa: E402  (the reference)

import oracle as O  # noqa: E402
from vidcom2_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}

CASES = [  # name, F, N, D, base, dtype, dist, seed
    ("cfg4", 512, 196, 3584, 0.25, "bf16", "drift", 0),
    ("cfg4", 512, 196, 3584, 0.25, "bf16", "iid", 1),
    ("target", 128, 196, 3584, 0.25, "bf16", "drift", 1),
    ("target", 128, 196, 3584, 0.25, "bf16", "drift", 2),
    ("target", 128, 196, 3584, 0.25, "bf16", "iid", 0),
    ("target", 128, 196, 3584, 0.25, "bf16", "iid", 1),
    ("target", 128, 196, 3584, 0.25, "bf16", "iid", 2),
    ("target", 128, 196, 3584, 0.25, "f16", "drift", 0),
    ("cfg3", 64, 324, 3584, 0.125, "bf16", "drift", 1),
    ("cfg3", 64, 324, 3584, 0.125, "bf16", "drift", 2),
    ("cfg3", 64, 324, 3584, 0.125, "bf16", "iid", 0),
    ("cfg3", 64, 324, 3584, 0.125, "bf16", "iid", 1),
    ("cfg3", 64, 324, 3584, 0.125, "bf16", "iid", 2),
    ("cfg5clip_full", 128, 196, 4096, 0.25, "f16", "drift", 0),
]


def main():
    out = []
    for name, F, N, D, base, dn, dist, seed in CASES:
        x = synth.make(F, N, D, DT[dn], seed, dist)
        t0 = time.time()
        sel = R.select_low_var_channels(x)
        v, f = R.compute_gaussian_scores(sel, N)
        s = -v.mean(dim=-1)
        scales = R.compute_scales(s, base)
        ks = (scales * N).round().long().clamp(min=1).tolist()
        idx = R.select_outlier_indices(v + f, scales, N)
        g = R._map_linear_offset(idx, N)
        rows = R.vidcom2_compression(x, model="qwen2_5_vl", base_scale=base, frame_token_len=N)
        dt_ref = time.time() - t0
        assert torch.equal(rows, x[g])
        O.set_mode("torch")
        o = O.compress_indices(x, N, base)
        rec = dict(name=name, F=F, N=N, D=D, base=base, dtype=dn, dist=dist, seed=seed, x_sha256=synth.sha256_tensor(x),
                   ks=ks, K=int(g.numel()), idx_sha256=synth.sha256_tensor(g), idx_head=g[:8].tolist(),
                   idx_tail=g[-8:].tolist(), out_sha256=synth.sha256_tensor(rows), v_sha256=synth.sha256_tensor(v),
                   f_sha256=synth.sha256_tensor(f), s=s.float().tolist(),
                   oracle_torch_mode=dict(ks=bool(o["ks"].tolist() == ks), idx=bool(torch.equal(o["global_idx"], g)),
                                          v_mismatch=int((o["v"].double() != v.double()).sum()),
                                          f_mismatch=int((o["f"].double() != f.double()).sum())))
        out.append(rec)
        print(name, dn, dist, seed, "K", rec["K"], f"reference {dt_ref:.2f}s", rec["oracle_torch_mode"], flush=True)
        del x, sel, v, f, rows
    with open(os.path.join(HERE, "big_cases.json"), "w") as fh:
        json.dump(dict(generator="tests/golden/make_big_golden.py", torch=torch.__version__,
                       threads=torch.get_num_threads(), cases=out), fh, indent=0)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
