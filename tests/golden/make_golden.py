#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

It imports token_compressor.vidcom2 from /root/reference, feeds it the bit-portable synthetic
inputs of vidcom2_amd/synth.py (regenerated from a seed at test time; only their sha256 is stored)
and records what the reference returns, stage by stage.  The fixtures are data only: inputs are
seeds, outputs are index lists / digests / a few sampled values.  Nothing of the reference's source
is copied.

Files written
  core_cases.json    full-pass goldens (channel order, scores digests + samples, budgets, kept indices)
  topk_kat.npz       torch.topk(largest=False) tie-breaking known-answer tests
  scales_kat.json    compute_scales / ks known-answer tests on hand-made frame scores
  misc_kat.json      exp tables (sha256 over all 65536 bit patterns), index mappers, error paths
"""
from __future__ import annotations

import hashlib
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

import oracle as O  # noqa: E402
from vidcom2_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def sha(t: torch.Tensor) -> str:
    return synth.sha256_tensor(t)


def ref_pass(x, tpf, base):
    """The reference's own stage functions, called exactly as vidcom2_compression does (vidcom2.py:27-36)."""
    var = x.var(dim=0, unbiased=False)
    _, cidx = torch.topk(var, k=int(x.shape[-1] * 0.5), largest=False)
    sel = R.select_low_var_channels(x)
    assert torch.equal(sel, x[:, cidx])
    v, f = R.compute_gaussian_scores(sel, tpf)
    s = -v.mean(dim=-1)
    scales = R.compute_scales(s, base)
    ks = (scales * tpf).round().long().clamp(min=1).tolist()
    idx = R.select_outlier_indices(v + f, scales, tpf)
    g = R._map_linear_offset(idx, tpf)
    return dict(var=var, chan_idx=cidx, v=v, f=f, s=s, scales=scales, ks=ks, global_idx=g)


def core_cases():
    shapes = [  # name, F, N, D, base, model, dtypes, dists, seeds
        ("toy", 4, 49, 64, 0.25, "qwen2_5_vl", ("f32", "bf16", "f16"), ("drift", "iid"), (0, 1, 2)),
        ("odd", 3, 50, 72, 0.3, "qwen2_vl", ("f32", "bf16", "f16"), ("drift",), (0, 1)),
        ("cfg1", 8, 196, 1024, 0.25, "llava_ov", ("f32", "bf16", "f16"), ("drift", "iid"), (0, 1, 2)),
        ("llava_vid", 16, 169, 3584, 0.15, "llava_vid", ("f32", "bf16", "f16"), ("drift", "iid"), (0, 1)),
        ("cfg2", 32, 196, 3584, 0.25, "llava_ov", ("f32", "bf16", "f16"), ("drift", "iid"), (0, 1, 2)),
        ("cfg3s", 16, 324, 3584, 0.125, "qwen2_5_vl", ("f32", "bf16", "f16"), ("drift", "iid"), (0, 1)),
        ("cfg3", 64, 324, 3584, 0.125, "qwen2_5_vl", ("bf16",), ("drift",), (0,)),
        ("target", 128, 196, 3584, 0.25, "llava_ov", ("bf16", "f32"), ("drift",), (0,)),
        ("cfg5clip", 16, 196, 4096, 0.25, "llava_ov", ("f16",), ("drift",), (0,)),
        ("lowk", 6, 196, 256, 0.01, "llava_ov", ("f32", "bf16"), ("drift",), (0,)),      # k*64<=N: partial_sort path
        ("highk", 6, 64, 128, 0.9, "qwen2_5_vl", ("f32", "bf16"), ("drift",), (0,)),     # clamp(max=1) path
        ("oneframe", 1, 196, 512, 0.25, "llava_ov", ("f32", "bf16"), ("iid",), (0,)),
    ]
    out = []
    for name, F, N, D, base, model, dts, dists, seeds in shapes:
        for dn in dts:
            for dist in dists:
                for seed in seeds:
                    x = synth.make(F, N, D, DT[dn], seed, dist)
                    r = ref_pass(x, N, base)
                    rows = x[r["global_idx"]]
                    o = O.compress_indices(x, N, base)
                    ov, of = o["v"], o["f"]
                    rec = dict(
                        name=name, F=F, N=N, D=D, base=base, model=model, dtype=dn, dist=dist, seed=seed,
                        x_sha256=sha(x), var_sha256=sha(r["var"]),
                        chan_idx=r["chan_idx"].tolist(), ks=r["ks"], global_idx=r["global_idx"].tolist(),
                        v_sha256=sha(r["v"]), f_sha256=sha(r["f"]), scales=r["scales"].float().tolist(),
                        s=r["s"].float().tolist(),
                        v_head=r["v"][0, :16].float().tolist(), f_head=r["f"][0, :16].float().tolist(),
                        v_mean=float(r["v"].double().mean()), f_mean=float(r["f"].double().mean()),
                        out_sha256=sha(rows), K=int(r["global_idx"].numel()),
                        # how the CPU oracle (exactly-rounded-op semantics) compares, measured here:
                        oracle=dict(
                            chan_idx=bool(torch.equal(o["chan_idx"], r["chan_idx"])),
                            ks=bool(o["ks"].tolist() == r["ks"]),
                            idx=bool(o["global_idx"].numel() == r["global_idx"].numel()
                                     and torch.equal(o["global_idx"], r["global_idx"])),
                            v_mismatch=int((ov.double() != r["v"].double()).sum()),
                            f_mismatch=int((of.double() != r["f"].double()).sum()),
                            max_abs_err=float(max((ov.double() - r["v"].double()).abs().max(),
                                                  (of.double() - r["f"].double()).abs().max())),
                        ),
                    )
                    rec["stable"] = rec["oracle"]["chan_idx"] and rec["oracle"]["ks"] and rec["oracle"]["idx"]
                    out.append(rec)
                    print(name, dn, dist, seed, "K", rec["K"], "stable", rec["stable"], rec["oracle"], flush=True)
    return out


def topk_kat():
    """torch.topk(largest=False) on tie-heavy vectors: both libstdc++ regimes, NaN, all-equal rows."""
    rng = np.random.RandomState(1234)
    vals, ks, sorteds, dtypes, offs_v, offs_o, outs = [], [], [], [], [0], [0], []
    cases = []
    for n in (5, 17, 64, 169, 196, 324, 1024, 3584, 4096):
        for levels in (1, 3, 8, 40, 0):           # number of distinct values (0 = continuous)
            for kfrac in (0.0, 0.01, 0.1, 0.25, 0.5, 0.9, 1.0):
                k = min(n, max(1, int(round(kfrac * n))))
                cases.append((n, levels, k))
    for ci, (n, levels, k) in enumerate(cases):
        if levels == 0:
            v = rng.randn(n).astype(np.float32)
        else:
            table = np.sort(rng.randn(levels).astype(np.float32))
            v = table[rng.randint(0, levels, size=n)]
        if ci % 7 == 3:
            v[rng.randint(0, n, size=max(1, n // 50))] = np.nan
        if ci % 11 == 5:
            v[rng.randint(0, n)] = -0.0
            v[rng.randint(0, n)] = 0.0
        dt = ("f32", "bf16", "f16")[ci % 3]
        t = torch.from_numpy(v).to(DT[dt])
        srt = bool(ci % 2)
        idx = torch.topk(t, k, largest=False, sorted=srt).indices
        assert torch.equal(O.topk_smallest(t, k, srt), idx), ("oracle topk mismatch", n, levels, k, dt, srt)
        vals.append(t.float().numpy())
        outs.append(idx.numpy().astype(np.int32))
        ks.append(k); sorteds.append(int(srt)); dtypes.append(("f32", "bf16", "f16").index(dt))
        offs_v.append(offs_v[-1] + n); offs_o.append(offs_o[-1] + k)
    np.savez_compressed(os.path.join(HERE, "topk_kat.npz"), values=np.concatenate(vals), out=np.concatenate(outs),
                        k=np.array(ks, np.int32), sorted=np.array(sorteds, np.int8), dtype=np.array(dtypes, np.int8),
                        offs_v=np.array(offs_v, np.int64), offs_o=np.array(offs_o, np.int64))
    print("topk KAT cases:", len(cases))


def scales_kat():
    vecs = {
        "peaky": [-1.0, -3.0, -3.1, -3.05, -2.9, -3.2, -3.0, -3.3],
        "flat": [-2.5] * 12,
        "two_tied_max": [-1.5, -1.5, -2.0, -2.5, -2.25, -3.0],
        "close": [-1.70, -1.71, -1.72, -1.705, -1.73, -1.74, -1.715, -1.75, -1.76, -1.77],
        "single": [-2.0],
        "wide": [-(0.5 + 0.13 * i) for i in range(64)],
    }
    out = []
    for name, v in vecs.items():
        for dn in ("f32", "bf16", "f16"):
            for base, tpf in ((0.25, 196), (0.125, 324), (0.9, 196), (1.0, 64), (0.001, 196), (0.15, 169)):
                s = torch.tensor(v, dtype=torch.float32).to(DT[dn])
                sc = R.compute_scales(s, base)
                ks = (sc * tpf).round().long().clamp(min=1).tolist()
                osc = O.compute_scales(s, base)
                out.append(dict(name=name, dtype=dn, base=base, tpf=tpf, s=s.float().tolist(),
                                scales=sc.float().tolist(), ks=ks,
                                oracle_equal=bool(torch.equal(osc, sc)),
                                oracle_ks_equal=bool(O.compute_ks(osc, tpf) == ks)))
    json.dump(out, open(os.path.join(HERE, "scales_kat.json"), "w"))
    print("scales KAT cases:", len(out), "oracle bit-equal:", sum(c["oracle_equal"] for c in out),
          "ks equal:", sum(c["oracle_ks_equal"] for c in out))


def misc_kat():
    out = {}
    # torch.exp over every bf16 / fp16 bit pattern (the argument of every exp on the path is a T value)
    exp = {}
    for dn in ("bf16", "f16"):
        bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
        t = bits.view(DT[dn])
        e = torch.exp(t)
        e = torch.where(e.isnan(), torch.full_like(e, float("nan")), e)   # canonical NaN payload
        exp[dn] = dict(sha256=hashlib.sha256(e.view(torch.int16).numpy().tobytes()).hexdigest(),
                       sample_in_bits=[0x3F80, 0xBF80, 0xC000, 0x0000, 0x4200, 0xC2B0],
                       sample_out_bits=[int(e.view(torch.int16)[b].item()) & 0xFFFF
                                        for b in (0x3F80, 0xBF80, 0xC000, 0x0000, 0x4200, 0xC2B0)])
    out["exp"] = exp
    # index mappers (vidcom2.py:99-115)
    idx = [torch.tensor([0, 12, 13, 168]), torch.tensor([5, 27, 100]), torch.tensor([168])]
    out["map_linear"] = dict(indices=[i.tolist() for i in idx], tpf=169,
                             out=R._map_linear_offset(idx, 169).tolist())
    out["map_grid_vid"] = dict(indices=[i.tolist() for i in idx], h=13, out=R._map_grid_vid(idx, 13).tolist())
    # error paths (exception type + message) and odd call forms
    errs = {}
    x = synth.make(2, 10, 8, torch.float32, 0, "iid")

    def grab(fn):
        try:
            fn()
            return None
        except Exception as e:  # noqa: BLE001
            return [type(e).__name__, str(e)]
    errs["unknown_model"] = grab(lambda: R.vidcom2_compression(x, model="nope"))
    errs["missing_tpf"] = grab(lambda: R.vidcom2_compression(x, model="qwen2_5_vl"))
    errs["missing_img"] = grab(lambda: R.vidcom2_compression(synth.make(2, 169, 8, torch.float32, 0, "iid"),
                                                             model="llava_vid"))
    errs["bad_rows"] = grab(lambda: R.vidcom2_compression(x, model="qwen2_vl", frame_token_len=7))
    out["errors"] = errs
    # frame_token_len given as a 1-element tensor (models/qwen2_vl.py:38-43)
    xq = synth.make(4, 25, 32, torch.bfloat16, 3, "drift")
    rq = R.vidcom2_compression(xq, model="qwen2_vl", frame_token_len=torch.tensor([25]))
    out["tensor_tpf"] = dict(F=4, N=25, D=32, dtype="bf16", seed=3, dist="drift", out_sha256=sha(rq),
                             shape=list(rq.shape))
    # llava_vid end to end: gathers rows of img_feat (grid with newline column), vidcom2.py:93-96
    F, h, D = 5, 13, 64
    flat = synth.make(F, h * h, D, torch.bfloat16, 7, "drift")
    img = synth.make(F, h * (h + 1), D, torch.bfloat16, 8, "iid")
    rv = R.vidcom2_compression(flat, model="llava_vid", base_scale=0.2, img_feat=img)
    out["llava_vid_e2e"] = dict(F=F, h=h, D=D, dtype="bf16", flat_seed=7, img_seed=8, base=0.2,
                                out_sha256=sha(rv), shape=list(rv.shape))
    json.dump(out, open(os.path.join(HERE, "misc_kat.json"), "w"))
    print("misc KATs written; errors:", errs)


def main():
    O.build()
    cases = core_cases()
    json.dump(dict(generator="tests/golden/make_golden.py", torch=torch.__version__, cases=cases),
              open(os.path.join(HERE, "core_cases.json"), "w"), separators=(",", ":"))
    n_st = sum(c["stable"] for c in cases)
    print(f"core cases: {len(cases)}, oracle index-exact on {n_st}")
    topk_kat()
    scales_kat()
    misc_kat()


if __name__ == "__main__":
    main()
