"""CPU: the oracle (oracle/vc2_oracle.cpp) against the golden vectors captured from the reference
itself (tests/golden/make_golden.py).  This is the pin that lets the oracle stand in for the
reference on the GPU box."""
import hashlib

import numpy as np
import pytest
import torch

import oracle as O
from vidcom2_amd import synth

from conftest import DT, case_id, load_core_cases, load_json, load_topk_kat, make_input

CASES = load_core_cases()
# the two 25k-token shapes cost ~15 s each to regenerate on CPU: keep one of each dtype there
CPU_CASES = [c for c in CASES if c["F"] * c["N"] * c["D"] <= 32 * 196 * 3584 or c["dtype"] == "bf16"]


KNOWN_RESIDUE = set()     # hinges on the fp32 order of the video-centre mean (not replayed)


@pytest.mark.parametrize("c", CPU_CASES, ids=case_id)
def test_torch_order_mode_is_bit_exact_to_reference(c):
    """Oracle mode 'torch' (replays torch's fp32 accumulation order of the norm and the row sums): scores,
    budgets and kept indices equal the reference's bit for bit in half precision."""
    if c["dtype"] == "f32":
        pytest.skip("fp32 is compared with a tolerance (test_full_pass_vs_reference)")
    O.set_mode("torch")
    try:
        x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
        o = O.compress_indices(x, c["N"], c["base"])
    finally:
        O.set_mode("exact")
    assert o["chan_idx"].tolist() == c["chan_idx"] and o["ks"].tolist() == c["ks"]
    if (c["name"], c["dtype"], c["dist"], c["seed"]) in KNOWN_RESIDUE:
        return
    assert synth.sha256_tensor(o["v"]) == c["v_sha256"] and synth.sha256_tensor(o["f"]) == c["f_sha256"]
    assert o["global_idx"].tolist() == c["global_idx"]


@pytest.mark.parametrize("c", CPU_CASES, ids=case_id)
def test_full_pass_vs_reference(c):
    O.set_mode("exact")
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
    assert synth.sha256_tensor(x) == c["x_sha256"], "synthetic generator is not bit-portable"
    o = O.compress_indices(x, c["N"], c["base"])
    _, var = O.select_low_var_channel_idx(x)
    # variance and channel ORDER (ascending variance, libstdc++ tie order): always bit-exact
    assert synth.sha256_tensor(var) == c["var_sha256"]
    assert o["chan_idx"].tolist() == c["chan_idx"]
    # scores: fp32 within 1e-5 of the reference (north_star tolerance); half types bit-exact except the
    # few accumulation-order-fragile values counted at generation time (DESIGN.md "Numerics contract")
    vh, fh = o["v"][0, :16].float().tolist(), o["f"][0, :16].float().tolist()
    if c["dtype"] == "f32":
        assert np.allclose(vh, c["v_head"], rtol=0, atol=1e-5) and np.allclose(fh, c["f_head"], rtol=0, atol=1e-5)
        assert abs(float(o["v"].double().mean()) - c["v_mean"]) < 1e-6
    elif c["oracle"]["v_mismatch"] == 0 and c["oracle"]["f_mismatch"] == 0:
        assert synth.sha256_tensor(o["v"]) == c["v_sha256"]
        assert synth.sha256_tensor(o["f"]) == c["f_sha256"]
    assert o["ks"].tolist() == c["ks"], "per-frame budgets differ from the reference"
    if c["stable"]:
        assert o["global_idx"].tolist() == c["global_idx"], "kept indices differ from the reference"
        assert synth.sha256_tensor(x[o["global_idx"]]) == c["out_sha256"]
    else:
        # reference scores within one rounding of an op boundary: indices may differ in a few frames
        a, b = set(o["global_idx"].tolist()), set(c["global_idx"])
        assert len(a) == len(b) and len(a & b) / len(b) > 0.95


def test_stable_fraction_documented():
    """All fp32 and all small/medium bf16 fixtures are index-exact; what is not is half precision at
    >= 5k tokens (fp16) / >= 20k tokens (bf16), where the reference's own fp32 accumulation order
    decides a rounding."""
    unstable = [c for c in CASES if not c["stable"]]
    assert all(c["dtype"] != "f32" for c in unstable)
    assert all(c["oracle"]["ks"] and c["oracle"]["chan_idx"] for c in CASES)
    assert all(c["F"] * c["N"] >= 2500 for c in unstable)
    assert len(unstable) <= 11


@pytest.mark.parametrize("i", range(0, 315, 1))
def test_topk_kat(i):
    v, k, srt, dn, want = TOPK[i]
    t = torch.from_numpy(v.copy()).to(DT[dn])
    assert O.topk_smallest(t, k, srt).tolist() == want.tolist()


TOPK = load_topk_kat()


def test_scales_kat():
    cases = load_json("scales_kat.json")
    n_equal = 0
    for c in cases:
        s = torch.tensor(c["s"], dtype=torch.float32).to(DT[c["dtype"]])
        sc = O.compute_scales(s, c["base"])
        want = torch.tensor(c["scales"], dtype=torch.float32)
        if c["dtype"] == "f32":
            assert torch.allclose(sc, want, rtol=0, atol=1e-6)
        n_equal += bool(torch.equal(sc.float(), want))
        assert O.compute_ks(sc, c["tpf"]) == c["ks"]
    assert n_equal >= len(cases) - 8     # fp32 softmax differs in the last bit on a few hand-made vectors


def test_exp_table():
    """RN_T(exp(x)) over every bf16 / fp16 bit pattern equals torch.exp's table (sha256 pinned)."""
    kat = load_json("misc_kat.json")["exp"]
    for dn in ("bf16", "f16"):
        bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
        x = bits.view(DT[dn])
        e = O.exp_T(x)
        e = torch.where(e.isnan(), torch.full_like(e, float("nan")), e)   # canonical NaN payload
        assert hashlib.sha256(e.view(torch.int16).numpy().tobytes()).hexdigest() == kat[dn]["sha256"]


def test_mappers_and_errors():
    kat = load_json("misc_kat.json")
    m = kat["map_linear"]
    assert O.map_linear_offset([torch.tensor(i) for i in m["indices"]], m["tpf"]).tolist() == m["out"]
    g = kat["map_grid_vid"]
    assert O.map_grid_vid([torch.tensor(i) for i in g["indices"]], g["h"]).tolist() == g["out"]
    x = synth.make(2, 10, 8, torch.float32, 0, "iid")
    with pytest.raises(ValueError, match="Unknown model: nope"):
        O.vidcom2_compression(x, model="nope")
    with pytest.raises(ValueError, match="frame_token_len required for qwen2_5_vl"):
        O.vidcom2_compression(x, model="qwen2_5_vl")
    with pytest.raises(ValueError, match="img_feat required for grid mapping"):
        O.vidcom2_compression(synth.make(2, 169, 8, torch.float32, 0, "iid"), model="llava_vid")
    with pytest.raises(RuntimeError):
        O.vidcom2_compression(x, model="qwen2_vl", frame_token_len=7)
    t = kat["tensor_tpf"]
    xq = synth.make(t["F"], t["N"], t["D"], DT[t["dtype"]], t["seed"], t["dist"])
    out = O.vidcom2_compression(xq, model="qwen2_vl", frame_token_len=torch.tensor([25]))
    assert list(out.shape) == t["shape"] and synth.sha256_tensor(out) == t["out_sha256"]
    v = kat["llava_vid_e2e"]
    flat = synth.make(v["F"], v["h"] * v["h"], v["D"], DT[v["dtype"]], v["flat_seed"], "drift")
    img = synth.make(v["F"], v["h"] * (v["h"] + 1), v["D"], DT[v["dtype"]], v["img_seed"], "iid")
    out = O.vidcom2_compression(flat, model="llava_vid", base_scale=v["base"], img_feat=img)
    assert list(out.shape) == v["shape"] and synth.sha256_tensor(out) == v["out_sha256"]


@pytest.mark.parametrize("c", [c for c in CASES if c["F"] * c["N"] * c["D"] <= 16 * 196 * 3584], ids=case_id)
def test_torch_restatement_equals_reference(c):
    """oracle/torch_restatement.py (the 'reference CPU path' leg of bench.py: the same aten ops in the same order,
    restated from the algorithm) reproduces the reference's fixtures bit for bit -- every dtype, fp32 included."""
    from oracle import torch_restatement as T
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
    r = T.compress(x, c["N"], c["base"])
    assert r["chan_idx"].tolist() == c["chan_idx"] and r["ks"] == c["ks"]
    assert r["global_idx"].tolist() == c["global_idx"]
    assert synth.sha256_tensor(r["v"]) == c["v_sha256"] and synth.sha256_tensor(r["f"]) == c["f_sha256"]
    assert synth.sha256_tensor(r["rows"]) == c["out_sha256"]


def test_oracle_matches_the_reference_on_adversarial_centre_means():
    """The `cancel` fixtures (make_adversarial_golden.py): torch's summation order decides centre-mean roundings there;
    the oracle's 'torch' mode reproduces the reference bit for bit, its 'exact' mode cannot on all of them."""
    import json
    import os
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "adversarial_cases.json")))["cases"]
    # (round 5) the same construction at the target shape and at cfg3, bf16 and fp16: 38 .. 335 frame centres per case
    # are decided by torch's summation order there
    cases += json.load(open(os.path.join(os.path.dirname(__file__), "golden", "adversarial_big_cases.json")))["cases"]
    assert sum(c["frame_centres_decided_by_order"] + c["video_centre_decided_by_order"] for c in cases) > 10
    assert sum(c["F"] * c["N"] * c["D"] >= 64 * 324 * 3584 for c in cases) >= 4
    O.set_mode("torch")
    try:
        for c in cases:
            x = synth.make(c["F"], c["N"], c["D"], DT[c["dtype"]], c["seed"], c["dist"])
            assert synth.sha256_tensor(x) == c["x_sha256"]
            r = O.compress_indices(x, c["N"], c["base"])
            assert r["ks"].tolist() == c["ks"] and r["global_idx"].tolist() == c["global_idx"]
            assert synth.sha256_tensor(r["v"]) == c["v_sha256"] and synth.sha256_tensor(r["f"]) == c["f_sha256"]
    finally:
        O.set_mode("exact")


def test_fp32_cancellation_residue_is_pinned():
    """VERDICT r5 item 5.  tests/golden/adversarial_f32_cases.json: the `cancel` inputs in fp32, through the reference.
    With fp32 scores nothing is rounded to a coarser type, so two tokens whose scores tie -- or differ in the last bit --
    are ordered by the last bit of torch's fp32 accumulation order and of its vectorised exp (MKL VML in this torch build:
    DESIGN.md section 3), which neither the oracle nor the kernels model.  What IS pinned, per case: budgets equal the
    reference's; scores within 1e-5 (measured < 5e-7); the kept set equals the reference's on the `stable` cases and, on
    the others, differs by exactly the recorded tokens (<= 0.2 % of the kept ones) whose REFERENCE scores lie within
    1e-6 of each other -- near-ties, not errors."""
    cases = load_json("adversarial_f32_cases.json")["cases"]
    assert len(cases) >= 6 and any(not c["torch"]["stable"] for c in cases) and any(c["torch"]["stable"] for c in cases)
    assert {(c["F"], c["N"], c["D"]) for c in cases} >= {(8, 196, 1024), (32, 196, 3584), (128, 196, 3584)}
    for c in cases:
        for mode in ("torch", "exact"):                  # (`exact`: fp64 accumulation -- what the HIP path does for fp32 inputs)
            m = c[mode]
            assert m["ks_equal"] and m["max_dv"] < 1e-5 and m["max_df"] < 1e-5
            # torch-order accumulation: <= 0.2 % of the kept tokens, reference scores within 5e-7; fp64 accumulation (the
            # kernels' fp32 path): <= 1 %, within 2e-6 -- 42 of 6272 at the target shape, the measured worst case
            assert m["tie_gap"] <= (5e-7 if mode == "torch" else 2e-6) and len(m["reference_only"]) == len(m["oracle_only"])
            assert len(m["reference_only"]) <= max(2, len(c["global_idx"]) // (500 if mode == "torch" else 100))
            assert m["stable"] == (not m["reference_only"])
    for c in [c for c in cases if c["F"] * c["N"] * c["D"] <= 16 * 196 * 1024]:     # (the larger ones: the -m gpu test, HIP == the `exact` relation)
        x = synth.make(c["F"], c["N"], c["D"], DT[c["dtype"]], c["seed"], c["dist"])
        assert synth.sha256_tensor(x) == c["x_sha256"]
        for mode in ("torch", "exact"):
            O.set_mode(mode)
            try:
                r = O.compress_indices(x, c["N"], c["base"])
            finally:
                O.set_mode("exact")
            assert r["ks"].tolist() == c["ks"]
            got = set(r["global_idx"].tolist())
            assert sorted(got - set(c["global_idx"])) == c[mode]["oracle_only"]
            assert sorted(set(c["global_idx"]) - got) == c[mode]["reference_only"]
            assert np.allclose(r["v"][0, :16].tolist(), c["v_head"], rtol=0, atol=1e-5)
            assert np.allclose(r["f"][0, :16].tolist(), c["f_head"], rtol=0, atol=1e-5)


def test_oracle_matches_the_reference_on_long_clips():
    """tests/golden/make_long_golden.py: more than 2^19 tokens per video, where torch's outer-sum cascade switches to
    level_power 5 (blocks of 32 rows).  Four of the fixtures are built (`cancel`) so that the reference's video centre
    IS the level-power-5 cascade's and differs from the level-power-4 one; the oracle's 'torch' mode reproduces the
    reference on all of them (indices, budgets, both score tensors)."""
    import json
    import os
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "long_cases.json")))["cases"]
    assert sum(c["video_centre"]["ne_level_power_4"] > 0 for c in cases) >= 4
    assert all(c["video_centre"]["ne_level_power_5"] == 0 and c["F"] * c["N"] > 1 << 19 for c in cases)
    O.set_mode("torch")
    try:
        for c in cases:
            x = synth.make(c["F"], c["N"], c["D"], DT[c["dtype"]], c["seed"], c["dist"])
            assert synth.sha256_tensor(x) == c["x_sha256"]
            r = O.compress_indices(x, c["N"], c["base"])
            assert synth.sha256_tensor(r["ks"].to(torch.int64)) == c["ks_sha256"]
            assert r["global_idx"].numel() == c["K"] and synth.sha256_tensor(r["global_idx"]) == c["idx_sha256"]
            assert synth.sha256_tensor(r["v"]) == c["v_sha256"] and synth.sha256_tensor(r["f"]) == c["f_sha256"]
    finally:
        O.set_mode("exact")


def test_cascade_level_power_host_rule():
    from vidcom2_amd.vidcom2 import cascade_level_power, cascade_modelled
    assert [cascade_level_power(n) for n in (1, 196, 1 << 19, (1 << 19) + 1, 1 << 23, (1 << 23) + 1, 1 << 27)] == \
        [4, 4, 4, 5, 5, 6, 6]
    assert cascade_modelled(3000 * 196) and cascade_modelled(1 << 25) and not cascade_modelled((1 << 25) + 4096)
