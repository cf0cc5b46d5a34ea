#!/usr/bin/env python
"""Adversarial fixtures for the centre means: the `cancel` inputs of vidcom2_amd/synth.py --
half-positive half-negative tokens, scored channels whose frame / video means are the residue of a ~100x cancellation -- run
through the REFERENCE itself (build container only: needs /root/reference):

    python tests/golden/make_adversarial_golden.py      ->  tests/golden/adversarial_cases.json
    python tests/golden/make_adversarial_golden.py big  ->  tests/golden/adversarial_big_cases.json (round 5: the same inputs at
                                                            the target shape and at cfg3, bf16 and fp16 -- VERDICT r4 item 4)
    python tests/golden/make_adversarial_golden.py f32  ->  tests/golden/adversarial_f32_cases.json (round 6, VERDICT r5 item 5:
                                                            the same inputs in fp32, where nothing is rounded to a coarser type
                                                            and a near-tie of two tokens' scores is decided by the last bit of
                                                            torch's fp32 accumulation order AND of its vectorised exp -- the
                                                            fixture records, per case, whether the oracle's `torch` mode picks
                                                            the reference's kept indices ("stable") and, where it does not, how
                                                            many tokens differ and how close the reference's own scores of the
                                                            swapped tokens are)

Data only (seeds, shapes, digests, index lists).  Also records, per case, how often the reference's centre values
differ from the exactly rounded means -- i.e. how often torch's fp32 cascade decided a rounding -- and the largest
distance (in fp32-ulps of the mean) between the exact sum and torch's: the quantity an ulp-of-the-mean margin cannot
bound."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from token_compressor.vidcom2 import vidcom2 as R  # noqa: E402  (the reference)
import torch.nn.functional as F  # noqa: E402

from vidcom2_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
CASES = [("adv", 16, 196, 512, "bf16", 0), ("adv", 16, 196, 512, "f16", 0), ("adv", 32, 100, 1024, "bf16", 1),
         ("adv", 8, 324, 768, "f16", 2), ("adv", 64, 196, 256, "bf16", 3)]
# (name, F, N, D, dtype, seed, base)
BIG_CASES = [("adv_target", 128, 196, 3584, "bf16", 4, 0.25), ("adv_target", 128, 196, 3584, "f16", 5, 0.25),
             ("adv_cfg3", 64, 324, 3584, "bf16", 6, 0.125), ("adv_cfg3", 64, 324, 3584, "f16", 7, 0.125)]


F32_CASES = [("adv_cfg1", 8, 196, 1024, "f32", 20, 0.25), ("adv_cfg1", 8, 196, 1024, "f32", 21, 0.25),
             ("adv_small", 16, 196, 512, "f32", 22, 0.25), ("adv_cfg2", 32, 196, 3584, "f32", 23, 0.25),
             ("adv_cfg2", 32, 196, 3584, "f32", 24, 0.25), ("adv_cfg3", 64, 324, 3584, "f32", 25, 0.125),
             ("adv_target", 128, 196, 3584, "f32", 26, 0.25), ("adv_target", 128, 196, 3584, "f32", 27, 0.25)]


def main_f32():
    """fp32 `cancel` inputs: reference outputs + how the oracle's `torch` mode relates to them (see the module docstring)."""
    import oracle as O
    out = []
    for (name, Fr, N, D, dn, seed, base) in F32_CASES:
        x = synth.make(Fr, N, D, DT[dn], seed, "cancel")
        sel = R.select_low_var_channels(x)
        v, f = R.compute_gaussian_scores(sel, N)
        scales = R.compute_scales(-v.mean(dim=-1), base)
        ks = (scales * N).round().long().clamp(min=1).tolist()
        idx = R.select_outlier_indices(v + f, scales, N)
        g = R._map_linear_offset(idx, N).tolist()
        total = (v + f).reshape(-1)
        rec = {"name": name, "F": Fr, "N": N, "D": D, "dtype": dn, "seed": seed, "dist": "cancel", "base": base,
               "x_sha256": synth.sha256_tensor(x), "ks": ks, "global_idx": g,
               "v_head": v[0, :16].tolist(), "f_head": f[0, :16].tolist(), "v_mean": float(v.double().mean())}
        # two relations: the oracle's `torch` mode (fp32 accumulation in torch's order: the closest restatement) and its
        # `exact` mode (fp64 accumulation: what the HIP path computes for fp32 inputs -- its kept set must equal THIS one)
        for mode in ("torch", "exact"):
            O.set_mode(mode)
            try:
                o = O.compress_indices(x, N, base)
            finally:
                O.set_mode("exact")
            og = o["global_idx"].tolist()
            only_ref, only_or = sorted(set(g) - set(og)), sorted(set(og) - set(g))
            # the reference's OWN scores of the tokens the two sides disagree on: a disagreement is a near-tie iff, frame by
            # frame, the scores of what one side keeps and the other drops are within the fp32 noise of each other
            gap = 0.0
            for fr in sorted({t // N for t in only_ref + only_or}):
                a = [float(total[t]) for t in only_ref if t // N == fr]
                b = [float(total[t]) for t in only_or if t // N == fr]
                gap = max(gap, max(a + b) - min(a + b))
            dv = float((o["v"].double() - v.double()).abs().max())
            df = float((o["f"].double() - f.double()).abs().max())
            rec[mode] = {"ks_equal": o["ks"].tolist() == ks, "stable": og == g, "oracle_only": only_or,
                         "reference_only": only_ref, "tie_gap": gap, "max_dv": dv, "max_df": df}
            print(name, Fr, N, D, seed, mode, "stable" if og == g else f"{len(only_ref)} of {len(g)} kept tokens differ",
                  f"tie gap {gap:.2e}  max|dv| {dv:.2e} max|df| {df:.2e}  ks equal: {o['ks'].tolist() == ks}")
        out.append(rec)
    json.dump({"cases": out}, open(os.path.join(HERE, "adversarial_f32_cases.json"), "w"), indent=None)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "f32":
        return main_f32()
    big = len(sys.argv) > 1 and sys.argv[1] == "big"
    out = []
    for case in (BIG_CASES if big else CASES):
        name, Fr, N, D, dn, seed = case[:6]
        base = case[6] if len(case) > 6 else 0.25
        x = synth.make(Fr, N, D, DT[dn], seed, "cancel")
        sel = R.select_low_var_channels(x)
        v, f = R.compute_gaussian_scores(sel, N)
        scales = R.compute_scales(-v.mean(dim=-1), base)
        ks = (scales * N).round().long().clamp(min=1).tolist()
        idx = R.select_outlier_indices(v + f, scales, N)
        g = R._map_linear_offset(idx, N)
        # diagnostics: the reference's centres against the exactly rounded means
        frames = F.normalize(sel.view(-1, N, sel.shape[-1]), dim=-1)
        fc_ref = frames.mean(dim=1)
        vc_ref = frames.mean(dim=(0, 1))
        fd = frames.double()
        fc_exact = (fd.sum(dim=1).float() / float(N)).to(DT[dn])
        vc_exact = (fd.sum(dim=(0, 1)).float() / float(Fr * N)).to(DT[dn])
        cancel = float((fd.abs().mean(dim=1) / fd.mean(dim=1).abs().clamp_min(1e-30)).median())
        out.append({"name": name, "F": Fr, "N": N, "D": D, "dtype": dn, "seed": seed, "dist": "cancel", "base": base,
                    "x_sha256": synth.sha256_tensor(x), "ks": ks, "global_idx": g.tolist(),
                    "v_sha256": synth.sha256_tensor(v), "f_sha256": synth.sha256_tensor(f),
                    "frame_centres_decided_by_order": int((fc_ref != fc_exact).sum()),
                    "video_centre_decided_by_order": int((vc_ref != vc_exact).sum()),
                    "median_cancellation": round(cancel, 1)})
        print(out[-1]["dtype"], Fr, N, D, "ks", ks[:6], "frame centres decided by torch's order:",
              out[-1]["frame_centres_decided_by_order"], "of", fc_ref.numel(), "| video:",
              out[-1]["video_centre_decided_by_order"], "of", vc_ref.numel(), "| cancellation x", cancel)
    json.dump({"cases": out}, open(os.path.join(HERE, "adversarial_big_cases.json" if big else "adversarial_cases.json"), "w"),
              indent=None if big else 1)


if __name__ == "__main__":
    main()
