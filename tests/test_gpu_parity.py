"""GPU (-m gpu): the HIP path, called through the C ABI, against (a) the CPU oracle on identical seeded
inputs and (b) the golden vectors captured from the reference.  Bar: kept indices, budgets, channel
selection bit-exact; half-precision scores bit-exact vs the oracle; fp32 scores within 1e-5."""
import ctypes
import os
import hashlib

import numpy as np
import pytest
import torch

import oracle as O
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
from vidcom2_amd._ffi import DTYPE_CODE, check, lib, ptr, stream_ptr

from conftest import DT, case_id, load_core_cases, load_json, load_topk_kat, make_input

pytestmark = pytest.mark.gpu
CASES = load_core_cases()
FP32_TOL = 1e-5          # north_star: "similarity scores within 1e-5 fp32"
# The one fixture whose reference result hinges on the fp32 accumulation order of the video-centre MEAN (not
# yet replayed; DESIGN.md "Numerics contract"): a single fp16 centre value flips, and with it a few v-scores.
KNOWN_RESIDUE = set()


@pytest.fixture(autouse=True)
def _default_mode():
    """Every test starts in the default ('torch' = bit-exact-to-reference) mode, oracle in the same mode."""
    _ffi.set_mode("torch")
    O.set_mode("torch")
    yield
    _ffi.set_mode("torch")
    O.set_mode("torch")


def dev():
    return torch.device("cuda:0")


def nan_eq(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


@pytest.mark.parametrize("mode", ["torch", "exact"])
@pytest.mark.parametrize("c", CASES, ids=case_id)
def test_full_pass(c, mode):
    """mode 'torch' (default): HIP == oracle(torch order) == the reference's golden vectors;
    mode 'exact': HIP == oracle(exact) (and == the reference wherever the reference is order-independent)."""
    _ffi.set_mode(mode)
    O.set_mode(mode)
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
    assert synth.sha256_tensor(x) == c["x_sha256"]
    xd = x.to(dev())
    keep = xd.clone()
    got = vc.compress(xd, c["N"], c["base"], want_scores=True)
    torch.cuda.synchronize()
    assert torch.equal(xd, keep), "input was modified"
    ref = O.compress_indices(x, c["N"], c["base"])
    gi = got.global_idx.cpu()
    # --- vs the oracle: everything bit-exact (fp32 scores: tolerance) ---
    assert got.ks.cpu().tolist() == ref["ks"].tolist()
    assert got.K == int(ref["global_idx"].numel()) and torch.equal(gi, ref["global_idx"])
    if c["dtype"] == "f32":
        assert (got.v_score.cpu() - ref["v"]).abs().max().item() <= 1e-6
        assert (got.f_score.cpu() - ref["f"]).abs().max().item() <= 1e-6
    else:
        assert nan_eq(got.v_score, ref["v"]) and nan_eq(got.f_score, ref["f"])
    assert torch.equal(got.rows.cpu(), x[gi])
    assert got.rows.dtype == x.dtype and gi.dtype == torch.int64
    # --- vs the reference's golden vectors ---
    assert got.ks.cpu().tolist() == c["ks"], "budgets differ from the reference"
    if c["dtype"] == "f32":
        assert np.allclose(got.v_score[0, :16].float().cpu().numpy(), c["v_head"], rtol=0, atol=FP32_TOL)
        assert np.allclose(got.f_score[0, :16].float().cpu().numpy(), c["f_head"], rtol=0, atol=FP32_TOL)
    key = (c["name"], c["dtype"], c["dist"], c["seed"])
    if mode == "torch" and key not in KNOWN_RESIDUE:
        # the headline claim: kept indices (and half-precision scores, bit for bit) equal the reference's
        assert gi.tolist() == c["global_idx"], "kept indices differ from the reference"
        assert synth.sha256_tensor(got.rows) == c["out_sha256"]
        if c["dtype"] != "f32":
            assert synth.sha256_tensor(got.v_score) == c["v_sha256"]
            assert synth.sha256_tensor(got.f_score) == c["f_sha256"]
    elif mode == "exact" and c["stable"]:
        assert gi.tolist() == c["global_idx"], "kept indices differ from the reference"
        assert synth.sha256_tensor(got.rows) == c["out_sha256"]


@pytest.mark.parametrize("c", [c for c in CASES if c["dtype"] != "f32"], ids=case_id)
def test_proven_centre_margins_agree_with_the_default(c):
    """Mode 3 decides which centre means to replay with a PROVEN bound on the cascade's error (relative to sum |x^|),
    the default with 16 ulps of the mean plus (frame means) 4 u A / n.  Same scores, budgets and kept indices on every
    half-precision fixture (and the end-to-end test of the bound's machinery: sweep-1 partials -> sum x^2, per-frame
    smallest denominator, the boundary-distance test).  The opt-in 'torch_fast' mode (the 16 ulps alone) agrees on these
    ordinary fixtures as well -- it is only the adversarial `cancel` inputs that it misses."""
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"]).to(dev())
    try:
        _ffi.set_mode("torch")
        a = vc.compress(x, c["N"], c["base"], want_scores=True)
        _ffi.set_mode("torch_proven")
        b = vc.compress(x, c["N"], c["base"], want_scores=True)
        _ffi.set_mode("torch_fast")
        f = vc.compress(x, c["N"], c["base"], want_scores=True)
    finally:
        _ffi.set_mode("torch")
    assert a.ks.tolist() == b.ks.tolist() and torch.equal(a.global_idx, b.global_idx)
    assert nan_eq(a.v_score, b.v_score) and nan_eq(a.f_score, b.f_score)
    assert a.ks.tolist() == f.ks.tolist() and torch.equal(a.global_idx, f.global_idx)
    assert nan_eq(a.v_score, f.v_score) and nan_eq(a.f_score, f.f_score)
    key = (c["name"], c["dtype"], c["dist"], c["seed"])
    if key not in KNOWN_RESIDUE:
        assert b.global_idx.cpu().tolist() == c["global_idx"] and synth.sha256_tensor(b.v_score) == c["v_sha256"]


def test_always_replay_equals_oracle():
    """Debug mode 2 replays torch's accumulation order for EVERY token: exercises the fix-up kernels on all rows."""
    O.set_mode("torch")
    for (F, N, D, dn, seed, dist) in [(4, 49, 64, "bf16", 0, "drift"), (8, 196, 1024, "bf16", 0, "drift"),
                                      (4, 100, 3584, "bf16", 0, "drift"), (4, 100, 3584, "f16", 0, "drift"),
                                      (4, 50, 4096, "bf16", 1, "iid"), (3, 40, 200, "bf16", 1, "iid"),
                                      (3, 50, 72, "f16", 1, "drift")]:
        x = make_input(F, N, D, dn, seed, dist)
        ref = O.compress_indices(x, N, 0.25)
        assert lib().vc2_set_mode(2) == 0
        got = vc.compress(x.to(dev()), N, 0.25, want_scores=True)
        assert nan_eq(got.v_score, ref["v"]) and nan_eq(got.f_score, ref["f"])
        assert torch.equal(got.global_idx.cpu(), ref["global_idx"])


@pytest.mark.parametrize("c", [c for c in CASES if c["name"] in ("toy", "odd", "cfg1", "llava_vid")], ids=case_id)
def test_stage_functions(c):
    """The reference's stage decomposition (select_low_var_channels ... map_features), one by one."""
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
    xd = x.to(dev())
    cidx = torch.tensor(c["chan_idx"])
    sel = vc.select_low_var_channels(xd)                       # column ORDER must match torch.topk's
    assert torch.equal(sel.cpu(), x[:, cidx])
    v, f = vc.compute_gaussian_scores(sel, c["N"])
    ov, of = O.compute_gaussian_scores(x[:, cidx], c["N"])
    if c["dtype"] == "f32":
        assert (v.cpu() - ov).abs().max() <= 1e-6 and (f.cpu() - of).abs().max() <= 1e-6
    else:
        assert nan_eq(v, ov) and nan_eq(f, of)
    # budgets from the oracle's scores so that the comparison isolates this stage
    s_o, tot_o = O.fuse(ov, of)
    scales = vc.compute_scales(s_o.to(dev()), c["base"])
    assert nan_eq(scales, O.compute_scales(s_o, c["base"]))
    idx = vc.select_outlier_indices(tot_o.to(dev()), scales, c["N"])
    want = O.select_outlier_indices(tot_o, O.compute_scales(s_o, c["base"]), c["N"])
    assert len(idx) == c["F"] and all(torch.equal(a.cpu(), b) for a, b in zip(idx, want))
    assert [i.numel() for i in idx] == c["ks"]
    g = vc._map_linear_offset(idx, c["N"])
    assert g.cpu().tolist() == c["global_idx"] or not c["stable"]
    rows = vc.map_features(idx, xd, None, vc.MODEL_SPECS["llava_ov"])
    assert torch.equal(rows.cpu(), x[g.cpu()])


def test_exp_table_on_device():
    kat = load_json("misc_kat.json")["exp"]
    for dn in ("bf16", "f16"):
        bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
        x = bits.view(DT[dn]).to(dev())
        out = torch.empty_like(x)
        check(lib().vc2_kat_exp(ptr(x), x.numel(), DTYPE_CODE[DT[dn]], ptr(out), stream_ptr(dev())), "kat_exp")
        e = out.cpu()
        e = torch.where(e.isnan(), torch.full_like(e, float("nan")), e)
        assert hashlib.sha256(e.view(torch.int16).numpy().tobytes()).hexdigest() == kat[dn]["sha256"]
        assert nan_eq(e, O.exp_T(bits.view(DT[dn])))


def test_rounding_instructions():
    """v_cvt_pk_bf16_f32 / v_cvt_f16_f32 round-trips equal c10's RNE conversions on edge values."""
    rng = np.random.RandomState(0)
    u = rng.randint(0, 2 ** 32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    edge = np.array([0x3F808000, 0x3F818000, 0x3F808001, 0x3F807FFF, 0x7F7FFFFF, 0x00000001, 0x80000001, 0x33000000,
                     0x33000001, 0x387FC000, 0x387FE000, 0x477FF000, 0x477FE000, 0x7F800000, 0xFF800000, 0x00800000],
                    dtype=np.uint32)
    f = torch.from_numpy(np.concatenate([u, edge]).view(np.float32).copy())
    f = f[~f.isnan()]
    fd = f.to(dev())
    for dn in ("bf16", "f16"):
        out = torch.empty(f.numel(), dtype=DT[dn], device=dev())
        check(lib().vc2_kat_round(ptr(fd), f.numel(), DTYPE_CODE[DT[dn]], ptr(out), stream_ptr(dev())), "kat_round")
        assert torch.equal(out.cpu(), f.to(DT[dn]))


TOPK = load_topk_kat()


@pytest.mark.parametrize("i", range(0, 315, 2))
def test_selection_kat(i):
    """The workgroup-parallel introselect replay gives torch.topk's SET on tie-heavy vectors -- both the
    channel-selection kernel and the per-frame selection kernel."""
    v, k, srt, dn, want = TOPK[i]
    n = v.size
    t = torch.from_numpy(v.copy()).to(DT[dn]).float().to(dev())
    want_set = sorted(want.tolist())
    mask = torch.zeros(n, dtype=torch.uint8, device=dev())
    cols = torch.full((n,), -1, dtype=torch.int32, device=dev())
    check(lib().vc2_chan_select(ptr(t), n, k, ptr(mask), ptr(cols), None, None, None, None, stream_ptr(dev())), "chan_select")
    assert mask.cpu().nonzero().flatten().tolist() == want_set
    assert cols[:k].cpu().tolist() == want_set and bool((cols[k:] == -1).all())
    if srt:     # torch.topk(sorted=True) ORDER, replayed on the device (introselect + introsort)
        order = torch.full((n,), -1, dtype=torch.int32, device=dev())
        opos = torch.full((n,), -1, dtype=torch.int32, device=dev())
        spos = torch.full((n,), -1, dtype=torch.int32, device=dev())
        perm = torch.full((n,), -1, dtype=torch.int32, device=dev())
        check(lib().vc2_chan_select(ptr(t), n, k, None, ptr(cols), ptr(perm), ptr(order), ptr(opos), ptr(spos),
                                    stream_ptr(dev())), "chan_select")
        assert sorted(perm[:k].cpu().tolist()) == want_set
        assert order[:k].cpu().tolist() == want.tolist()
        assert cols[opos[:k].long()].cpu().tolist() == want.tolist()
        assert order[spos[:k].long()].cpu().tolist() == cols[:k].cpu().tolist()
    # per-frame path: 3 identical frames with scale = k/n  ->  ks = k
    if n >= 17:
        F = 3
        scores = t.repeat(F, 1).contiguous()
        scales = torch.full((F,), k / n, dtype=torch.float32, device=dev())
        idx = vc.select_outlier_indices(scores, scales, n)
        assert all(a.cpu().tolist() == want_set for a in idx)


def test_scales_kat():
    for c in load_json("scales_kat.json"):
        s = torch.tensor(c["s"], dtype=torch.float32).to(DT[c["dtype"]])
        sc = vc.compute_scales(s.to(dev()), c["base"])
        assert nan_eq(sc, O.compute_scales(s, c["base"]))
        if c["dtype"] == "f32":
            assert torch.allclose(sc.cpu(), torch.tensor(c["scales"]), rtol=0, atol=1e-6)
        elif c["oracle_equal"]:
            assert sc.float().cpu().tolist() == c["scales"]
        N = c["tpf"]
        idx = vc.select_outlier_indices(torch.zeros(len(c["s"]), N, dtype=DT[c["dtype"]], device=dev()), sc, N)
        assert [i.numel() for i in idx] == c["ks"]


def test_mappers_errors_and_call_forms():
    kat = load_json("misc_kat.json")
    m, g = kat["map_linear"], kat["map_grid_vid"]
    ind = [torch.tensor(i, device=dev()) for i in m["indices"]]
    assert vc._map_linear_offset(ind, m["tpf"]).cpu().tolist() == m["out"]
    assert vc._map_grid_vid(ind, g["h"]).cpu().tolist() == g["out"]
    x = synth.make(2, 10, 8, torch.float32, 0, "iid").to(dev())
    with pytest.raises(RuntimeError):
        vc.vidcom2_compression(x, model="qwen2_vl", frame_token_len=7)       # 20 rows % 7 != 0
    t = kat["tensor_tpf"]
    xq = synth.make(t["F"], t["N"], t["D"], DT[t["dtype"]], t["seed"], t["dist"]).to(dev())
    out = vc.vidcom2_compression(xq, model="qwen2_vl", frame_token_len=torch.tensor([25]))
    assert list(out.shape) == t["shape"] and synth.sha256_tensor(out) == t["out_sha256"]
    v = kat["llava_vid_e2e"]
    flat = synth.make(v["F"], v["h"] * v["h"], v["D"], DT[v["dtype"]], v["flat_seed"], "drift").to(dev())
    img = synth.make(v["F"], v["h"] * (v["h"] + 1), v["D"], DT[v["dtype"]], v["img_seed"], "iid").to(dev())
    out = vc.vidcom2_compression(flat, model="llava_vid", base_scale=v["base"], img_feat=img)
    assert list(out.shape) == v["shape"] and synth.sha256_tensor(out) == v["out_sha256"]


def test_degenerate_inputs():
    # constant input: every variance is exactly 0, every score ties -> pure tie-breaking
    for dn in ("f32", "bf16"):
        x = torch.full((4 * 49, 64), 0.5, dtype=DT[dn])
        got = vc.compress(x.to(dev()), 49, 0.25)
        ref = O.compress_indices(x, 49, 0.25)
        assert got.ks.cpu().tolist() == ref["ks"].tolist() and torch.equal(got.global_idx.cpu(), ref["global_idx"])
    # all-zero rows: fp16 gives NaN scores (eps underflows), bf16/fp32 give finite ones
    for dn in ("f16", "bf16"):
        x = synth.make(3, 49, 64, DT[dn], 0, "iid")
        x[5] = 0
        x[60] = 0
        got = vc.compress(x.to(dev()), 49, 0.25, want_scores=True)
        ref = O.compress_indices(x, 49, 0.25)
        assert nan_eq(got.v_score, ref["v"])
        assert got.ks.cpu().tolist() == ref["ks"].tolist() and torch.equal(got.global_idx.cpu(), ref["global_idx"])


@pytest.mark.parametrize("shape", [(128, 196, 3584, "bf16", 0.25), (64, 324, 3584, "bf16", 0.125),
                                   (128, 196, 4096, "f16", 0.25), (128, 196, 3584, "f32", 0.25)],
                         ids=lambda s: "x".join(map(str, s)))
def test_full_size_properties(shape):
    """BASELINE.json's full sizes through size-independent properties (the oracle comparison at these
    sizes is in test_full_pass for the fixtures that exist)."""
    F, N, D, dn, base = shape
    x = make_input(F, N, D, dn, 0, "drift").to(dev())
    a = vc.compress(x, N, base)
    b = vc.compress(x, N, base)
    assert torch.equal(a.global_idx, b.global_idx) and torch.equal(a.ks, b.ks)          # deterministic
    ks = a.ks.cpu()
    gi = a.global_idx.cpu()
    assert a.K == int(ks.sum()) and (ks >= 1).all() and (ks <= N).all()
    assert abs(a.K - base * F * N) <= 0.02 * base * F * N + F                             # budget conserved
    assert bool((gi[1:] > gi[:-1]).all())                                                  # globally ascending
    assert torch.equal(torch.bincount(gi // N, minlength=F), ks)                           # k_f per frame
    assert torch.equal(a.rows, x[a.global_idx])                                            # gather = index
    # scaling by a power of two is exact in every dtype: identical selection
    c = vc.compress(x * 2, N, base)
    assert torch.equal(c.global_idx, a.global_idx) and torch.equal(c.ks, a.ks)
    # a second, different video concatenated then split: frames of the first half keep their own ranking
    # only relative to the joint centre -> just check the concatenation runs and budgets still add up
    if F <= 64:
        y = torch.cat([x, x.flip(0)])
        d = vc.compress(y, N, base)
        assert d.K == int(d.ks.sum())


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_integration_md_ctypes_stub_runs():
    """The ctypes stub printed in INTEGRATION.md §2 is executed as written (only the library path is filled in)
    and must reproduce the package's own result."""
    import re
    from vidcom2_amd import vidcom2 as V
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\nimport ctypes, torch\n(.*?)```", text, re.S)
    assert block, "INTEGRATION.md lost its ctypes stub"
    src = "import ctypes, torch\n" + block.group(1)
    src = src.replace('".../vidcom2_amd/_lib/libvc2hip.so"', repr(_ffi.LIB_PATH))
    ns = {}
    exec(compile(src, "INTEGRATION.md", "exec"), ns)
    x = make_input(8, 196, 1024, "bf16", 3, "drift").cuda()
    rows, idx, ks = ns["compress"](x, 196, 0.25)
    ref = V.compress(x, 196, 0.25)
    assert torch.equal(idx, ref.global_idx) and torch.equal(ks, ref.ks) and torch.equal(rows, ref.rows)


def test_pass_is_hip_graph_capturable():
    """The whole pass only enqueues (no host sync, no allocation): it can be captured into a hipGraph -- side-stream
    fork/join included -- and replayed (serving loops)."""
    x = make_input(32, 196, 1024, "bf16", 5, "drift").cuda()
    plan = vc.vidcom2.CompressPlan(32, 196, 1024, torch.bfloat16, x.device, 0.25)
    plan.enqueue(x)
    ref = plan.finish()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.enqueue(x)                                   # warm-up on the capture stream (lazy side-stream creation)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        plan.enqueue(x)
    plan.idx.zero_()
    g.replay()
    got = plan.finish()
    assert torch.equal(got.global_idx, ref.global_idx) and torch.equal(got.ks, ref.ks) and torch.equal(got.rows, ref.rows)
    x2 = make_input(32, 196, 1024, "bf16", 6, "drift").cuda()
    x.copy_(x2)                                           # same buffers, new contents
    g.replay()
    got2 = plan.finish()
    want2 = vc.vidcom2.compress(x2, 196, 0.25)
    assert torch.equal(got2.global_idx, want2.global_idx) and torch.equal(got2.rows, want2.rows)


@pytest.mark.parametrize("F", [1024, 1100])
def test_many_frames_budget_paths(F):
    """F <= 1024: every k_select wave derives the budgets itself; above: the separate k_scales kernel."""
    x = make_input(F, 16, 64, "bf16", 11, "drift")
    got = vc.vidcom2.compress(x.cuda(), 16, 0.25)
    ref = O.compress_indices(x, 16, 0.25)
    assert torch.equal(got.ks.cpu(), ref["ks"]) and torch.equal(got.global_idx.cpu(), ref["global_idx"])


def test_two_clips_in_flight_on_two_streams():
    """Every caller stream gets its own internal side stream: two passes enqueued back to back on two streams run
    concurrently and still produce what they produce alone."""
    xs = [make_input(32, 196, 1024, "bf16", s, "drift").cuda() for s in (21, 22)]
    want = [vc.vidcom2.compress(x, 196, 0.25) for x in xs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    plans = [vc.vidcom2.CompressPlan(32, 196, 1024, torch.bfloat16, xs[0].device, 0.25) for _ in xs]
    torch.cuda.synchronize()
    for _ in range(5):
        for st, pl, x in zip(streams, plans, xs):
            with torch.cuda.stream(st):
                pl.enqueue(x)
    got = []
    for st, pl in zip(streams, plans):
        with torch.cuda.stream(st):
            got.append(pl.finish())
    for g, w in zip(got, want):
        assert torch.equal(g.global_idx, w.global_idx) and torch.equal(g.ks, w.ks) and torch.equal(g.rows, w.rows)


def test_compress_batch_matches_sequential():
    shapes = [(16, 196, 1024), (8, 196, 1024), (16, 196, 1024), (4, 196, 1024), (12, 196, 1024)]
    xs = [make_input(F, N, D, "bf16", 30 + i, "drift").cuda() for i, (F, N, D) in enumerate(shapes)]
    want = [vc.vidcom2.compress(x, 196, 0.25) for x in xs]
    for lanes in (1, 2, 3):
        got = vc.compress_batch(xs, 196, 0.25, in_flight=lanes)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g.K == w.K and torch.equal(g.global_idx, w.global_idx) and torch.equal(g.rows, w.rows)
    assert vc.compress_batch([], 196) == []
    with pytest.raises(RuntimeError):
        vc.compress_batch([xs[0][:100]], 196)


def test_compress_batch_same_shape_reuses_lanes_and_plans():
    """A batch of clips of one shape runs on compress_batch's kept lane streams with one cached plan per lane: a second
    call creates neither streams nor plans (the plan cache is keyed by stream -- fresh streams per call once meant fresh
    plans and workspaces per call), and every clip still gets what it gets alone."""
    from vidcom2_amd import vidcom2 as V
    xs = [make_input(16, 196, 1024, "f16", 70 + i, "drift").cuda() for i in range(7)]
    want = [V.compress(x, 196, 0.25) for x in xs]
    V.compress_batch(xs, 196, 0.25)
    dev = xs[0].device
    lanes_before = list(V._lane_streams(dev, V.BATCH_LANES - 1))
    plans_before = dict(V._PLAN_CACHE)
    got = V.compress_batch(xs, 196, 0.25)
    assert [s.cuda_stream for s in V._lane_streams(dev, V.BATCH_LANES - 1)] == [s.cuda_stream for s in lanes_before]
    assert set(V._PLAN_CACHE.keys()) == set(plans_before.keys())
    assert all(V._PLAN_CACHE[k] is plans_before[k] for k in plans_before)
    for g, w in zip(got, want):
        assert g.K == w.K and torch.equal(g.global_idx, w.global_idx) and torch.equal(g.rows, w.rows) and torch.equal(g.ks, w.ks)
    # the cached plans do not keep pointing into the batch's buffers
    for p in V._PLAN_CACHE.values():
        assert p.rows is None or p.rows.data_ptr() != got[0].rows.data_ptr()


def test_selection_engine_loop_bounds_never_expire():
    """Every loop of the selection / sort replay (vc2_select2.h) is bounded; a bound that expires is counted on the
    device.  After a spread of passes (incl. everything the tests above ran in this process) all counters are 0."""
    for (F, N, D, dn) in [(1, 196, 512, "bf16"), (6, 196, 256, "bf16"), (8, 196, 1024, "f16"), (4, 100, 3584, "bf16"),
                          (3, 1100, 128, "bf16"), (2, 2100, 64, "f32")]:
        x = make_input(F, N, D, dn, 0, "iid")
        vc.compress(x.to(dev()), N, 0.25)
        vc.vidcom2.low_var_channel_order(x.to(dev()))
    out = (ctypes.c_int32 * 8)()
    check(lib().vc2_selftest_counters(out, 0), "vc2_selftest_counters")
    assert list(out) == [0] * 8, list(out)


def test_select_with_tpf_not_equal_to_row_length():
    """vidcom2.py:72 multiplies the scales by `tpf`, not by scores.shape[1]: smaller tpf -> smaller budgets; a budget
    beyond the row length is torch.topk's 'k out of range' RuntimeError."""
    F, N = 5, 120
    g = torch.Generator().manual_seed(5)
    scores = torch.randn(F, N, generator=g).to(torch.bfloat16)
    scales = torch.tensor([0.25, 0.5, 0.1, 0.3, 0.9]).to(torch.bfloat16)
    for tpf in (60, 120, 130):
        ks = (scales * tpf).round().long().clamp(min=1).tolist()          # the reference's own line
        want = [torch.topk(scores[i], k, largest=False, sorted=False).indices.sort().values for i, k in enumerate(ks)]
        got = vc.select_outlier_indices(scores.to(dev()), scales.to(dev()), tpf)
        assert [g_.cpu().tolist() for g_ in got] == [w.tolist() for w in want]
    with pytest.raises(RuntimeError, match="out of range"):
        vc.select_outlier_indices(scores.to(dev()), scales.to(dev()), 200)     # 0.9 * 200 = 180 > 120


def test_multi_scale_gaussian_broadcasts_like_torch():
    """(x - center) broadcasts in the reference helper (vidcom2.py:61): per-token centres, 2-D x, scalar-row centre."""
    from vidcom2_amd.vidcom2 import _multi_scale_gaussian
    g = torch.Generator().manual_seed(9)
    alphas = [2 ** k for k in range(-3, 2)]

    def ref(x, c):        # the reference's two lines, on CPU in T
        d = ((x - c) ** 2).sum(-1)
        return sum(torch.exp(-d / (2 * a)) for a in alphas)

    for dt in (torch.bfloat16, torch.float32):
        x = (torch.randn(3, 7, 96, generator=g) * 0.1).to(dt)
        for c in [(torch.randn(3, 7, 96, generator=g) * 0.1).to(dt), (torch.randn(1, 1, 96, generator=g) * 0.1).to(dt),
                  (torch.randn(3, 1, 96, generator=g) * 0.1).to(dt), (torch.randn(96, generator=g) * 0.1).to(dt),
                  (torch.randn(1, 7, 96, generator=g) * 0.1).to(dt)]:
            got = _multi_scale_gaussian(x.to(dev()), c.to(dev()), alphas).cpu()
            want = ref(x, c)
            assert got.shape == want.shape
            if dt == torch.float32:
                assert (got - want).abs().max() <= 1e-5
            else:
                assert torch.equal(got, want)
        x2 = (torch.randn(11, 96, generator=g) * 0.1).to(dt)
        c2 = (torch.randn(96, generator=g) * 0.1).to(dt)
        got = _multi_scale_gaussian(x2.to(dev()), c2.to(dev()), alphas).cpu()
        assert got.shape == (11,) and ((got.float() - ref(x2, c2).float()).abs().max() <= (1e-5 if dt == torch.float32 else 0))


def test_replay_everything_beyond_the_old_queue_sizes():
    """Debug mode 2 replays torch's order for EVERY token: 2 * 33320 sums here, more than the 65536-entry queue the
    first version dropped silently (the queues are gone: sweep 3 replays inside its workgroups, and the remaining
    lists hold one entry per row / per (frame, column) pair, so nothing can overflow)."""
    F, N, D = 170, 196, 64
    x = make_input(F, N, D, "bf16", 4, "drift")
    O.set_mode("torch")
    ref = O.compress_indices(x, N, 0.25)
    assert lib().vc2_set_mode(2) == 0
    got = vc.compress(x.to(dev()), N, 0.25, want_scores=True)
    assert nan_eq(got.v_score, ref["v"]) and nan_eq(got.f_score, ref["f"])
    assert torch.equal(got.global_idx.cpu(), ref["global_idx"])


def test_gather_source_too_small_raises_like_the_reference():
    x = make_input(4, 169, 128, "bf16", 0, "drift").to(dev())
    with pytest.raises(IndexError):
        vc.vidcom2_compression(x, model="llava_vid", img_feat=x[:100])       # needs 4 * 13 * 14 rows


@pytest.mark.gpu
def test_plan_cache_returns_independent_results():
    """The one-shot API re-uses a cached plan (workspace, status words) per shape / stream / thread, but every call
    hands out fresh tensors, like the reference: a second call must not overwrite the first call's result."""
    from vidcom2_amd import vidcom2 as V
    V.clear_plan_cache()
    xa = synth.make(8, 49, 256, torch.bfloat16, 1, "drift").cuda()
    xb = synth.make(8, 49, 256, torch.bfloat16, 2, "drift").cuda()
    ra = V.compress(xa, 49, 0.25)
    rows_a, idx_a, ks_a = ra.rows.clone(), ra.global_idx.clone(), ra.ks.clone()
    n_plans = len(V._PLAN_CACHE)
    rb = V.compress(xb, 49, 0.25)
    assert len(V._PLAN_CACHE) == n_plans == 1                       # same shape, stream, thread: one plan
    assert torch.equal(ra.rows, rows_a) and torch.equal(ra.global_idx, idx_a) and torch.equal(ra.ks, ks_a)
    assert ra.rows.data_ptr() != rb.rows.data_ptr() and ra.global_idx.data_ptr() != rb.global_idx.data_ptr()
    assert not torch.equal(rb.global_idx, idx_a) or not torch.equal(rb.rows, rows_a)
    # the same input again gives the same answer from the re-used workspace
    rc = V.compress(xa, 49, 0.25)
    assert torch.equal(rc.rows, rows_a) and torch.equal(rc.global_idx, idx_a)
    # another stream gets its own plan
    with torch.cuda.stream(torch.cuda.Stream()):
        rd = V.compress(xa, 49, 0.25)
    torch.cuda.synchronize()
    assert len(V._PLAN_CACHE) == 2 and torch.equal(rd.global_idx, idx_a)


@pytest.mark.gpu
def test_mode_is_process_wide_with_a_per_thread_override():
    """vc2_set_mode applies to passes issued from ANY thread (HF generate streams from a worker thread);
    vc2_set_thread_mode overrides it for one thread without disturbing the others."""
    import threading
    seen = {}

    def worker(name, override):
        if override is not None:
            _ffi.set_thread_mode(override)
        seen[name] = _ffi.get_mode()

    try:
        _ffi.set_mode("exact")
        t = threading.Thread(target=worker, args=("follows", None)); t.start(); t.join()
        t = threading.Thread(target=worker, args=("own", "torch")); t.start(); t.join()
        assert seen == {"follows": "exact", "own": "torch"} and _ffi.get_mode() == "exact"
        _ffi.set_thread_mode("torch")
        assert _ffi.get_mode() == "torch"
        t = threading.Thread(target=worker, args=("other", None)); t.start(); t.join()
        assert seen["other"] == "exact"                              # this thread's override is its own
        _ffi.set_mode("torch")                                       # process-wide; drops the caller's override
        assert _ffi.get_mode() == "torch"
    finally:
        _ffi.set_mode("torch")
        _ffi.set_thread_mode(None)


ADV = load_json("adversarial_cases.json")["cases"] + load_json("adversarial_big_cases.json")["cases"]   # (round 5: + target / cfg3 size)


@pytest.mark.parametrize("mode", ["torch", "torch_proven"])
@pytest.mark.parametrize("c", ADV, ids=lambda c: f"adv-{c['F']}x{c['N']}x{c['D']}-{c['dtype']}")
def test_adversarial_centre_means(c, mode):
    """tests/golden/make_adversarial_golden.py: inputs BUILT so that torch's fp32 cascade decides centre-mean roundings
    (tiny addends meet a large running sum that later cancels: `frame_centres_decided_by_order` > 0 in the fixture)
    and the cascade's error is hundreds of ulps of the mean.  Scores, budgets and kept indices must equal the
    reference's in the default mode (since round 4 the frame means' replay margin has a term relative to sum |x^|: no
    exception left) and in the proven-margin mode."""
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
    assert synth.sha256_tensor(x) == c["x_sha256"]
    try:
        _ffi.set_mode(mode)
        got = vc.compress(x.to(dev()), c["N"], c["base"], want_scores=True)
    finally:
        _ffi.set_mode("torch")
    assert got.ks.cpu().tolist() == c["ks"]
    assert got.global_idx.cpu().tolist() == c["global_idx"]
    assert synth.sha256_tensor(got.v_score) == c["v_sha256"] and synth.sha256_tensor(got.f_score) == c["f_sha256"]


@pytest.mark.parametrize("c", load_json("adversarial_f32_cases.json")["cases"],
                         ids=lambda c: f"advf32-{c['F']}x{c['N']}x{c['D']}-{c['seed']}")
def test_fp32_cancellation_residue(c):
    """the `cancel` inputs in fp32 (tests/golden/adversarial_f32_cases.json, from the reference).  The HIP
    path must stand where the oracle's fp64-accumulating mode stands (what the kernels do for fp32 inputs): budgets equal to
    the reference's, scores within 1e-5, the reference's kept indices on the `stable` cases and, on the others, exactly the
    recorded near-tie tokens swapped (reference scores within 1e-6 of each other: decided by torch's fp32 summation order
    and its vectorised exp)."""
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
    assert synth.sha256_tensor(x) == c["x_sha256"]
    got = vc.compress(x.to(dev()), c["N"], c["base"], want_scores=True)
    assert got.ks.cpu().tolist() == c["ks"]
    kept = set(got.global_idx.cpu().tolist())
    m = c["exact"]                   # (fp32 inputs: the kernels accumulate in fp64 -- the oracle's `exact` relation to the reference)
    assert sorted(kept - set(c["global_idx"])) == m["oracle_only"]
    assert sorted(set(c["global_idx"]) - kept) == m["reference_only"]
    assert np.allclose(got.v_score.cpu()[0, :16].tolist(), c["v_head"], rtol=0, atol=1e-5)
    assert np.allclose(got.f_score.cpu()[0, :16].tolist(), c["f_head"], rtol=0, atol=1e-5)
    assert abs(float(got.v_score.double().mean()) - c["v_mean"]) < 1e-6


@pytest.mark.parametrize("ord_form", ["1", "0"])
@pytest.mark.parametrize("c", [c for c in ADV if c["D"] in (1024, 3584)], ids=lambda c: f"ord-{c['F']}x{c['N']}x{c['D']}-{c['dtype']}")
def test_torch_ordered_frame_sums_reproduce_the_reference(c, ord_form):
    """VC2_S2_ORD=1 (the DEFAULT since round 6): sweep 2 adds every frame's x^ in torch's own order (16-row blocks,
    k_norm_colsum2<.., ORD>) and k_frame_centres combines the block sums the way torch's cascade does -- no margin, no
    replay for any frame mean.  VC2_S2_ORD=0: the form of rounds 1-5 (exact fp64 frame sums, a margin around every T
    rounding boundary, boundary-near means replayed in torch's order).  Two completely different routes to the same bits:
    on the adversarial `cancel` inputs (the reference's own centres are decided by the summation order in up to 335
    places) BOTH must reproduce the reference exactly."""
    import os
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
    old = os.environ.get("VC2_S2_ORD")
    try:
        os.environ["VC2_S2_ORD"] = ord_form
        got = vc.compress(x.to(dev()), c["N"], c["base"], want_scores=True)
        torch.cuda.synchronize()
    finally:
        if old is None:
            del os.environ["VC2_S2_ORD"]
        else:
            os.environ["VC2_S2_ORD"] = old
    assert got.ks.cpu().tolist() == c["ks"]
    assert got.global_idx.cpu().tolist() == c["global_idx"]
    assert synth.sha256_tensor(got.v_score) == c["v_sha256"] and synth.sha256_tensor(got.f_score) == c["f_sha256"]


@pytest.mark.parametrize("shape", [(5, 169, 1024, "bf16", "drift"), (3, 100, 1024, "f16", "cancel"), (9, 324, 1024, "bf16", "iid"),
                                   (4, 16, 1024, "f16", "drift"), (6, 7, 1024, "bf16", "cancel"), (2, 512, 1024, "f16", "iid"),
                                   (40, 33, 1024, "bf16", "cancel"), (130, 48, 1024, "f16", "drift")],
                         ids=lambda s_: "x".join(map(str, s_)))
@pytest.mark.parametrize("ord_form", ["1", "0"])
def test_torch_ordered_frame_sums_on_odd_frame_lengths(shape, ord_form):
    """The ORD form (and, ord_form = 0, the margin-and-replay form it replaced as the default) against the oracle where the
    ORD geometry has corners: frames of 9 / 4 leftover rows, none, fewer than one block, 32 blocks, more blocks than a
    workgroup's waves, unequal pieces per frame (OrdGeo), several blocks per wave (many frames)."""
    import os
    F, N, D, dn, dist = shape
    x = make_input(F, N, D, dn, 11, dist)
    O.set_mode("torch")
    ref = O.compress_indices(x, N, 0.25)
    old = os.environ.get("VC2_S2_ORD")
    try:
        os.environ["VC2_S2_ORD"] = ord_form
        got = vc.compress(x.to(dev()), N, 0.25, want_scores=True)
        torch.cuda.synchronize()
    finally:
        if old is None:
            del os.environ["VC2_S2_ORD"]
        else:
            os.environ["VC2_S2_ORD"] = old
    assert nan_eq(got.v_score, ref["v"]) and nan_eq(got.f_score, ref["f"])
    assert got.ks.cpu().tolist() == ref["ks"].tolist()
    assert torch.equal(got.global_idx.cpu(), ref["global_idx"])


@pytest.mark.parametrize("shape", [(1024, 16, 256, "bf16"), (300, 7, 128, "f16"), (640, 40, 512, "bf16"), (97, 33, 192, "f16"),
                                   (2000, 4, 64, "bf16")], ids=lambda s_: "x".join(map(str, s_)))
def test_many_short_frames(shape):
    """Sweep 2 cuts the rows into equal chunks whatever the frames are: with short frames a chunk holds several whole
    frames and pieces of two more (one flush of the column sums per segment).  Scores, budgets and kept indices
    against the oracle."""
    F, N, D, dn = shape
    x = make_input(F, N, D, dn, 5, "drift")
    O.set_mode("torch")
    ref = O.compress_indices(x, N, 0.25)
    got = vc.compress(x.to(dev()), N, 0.25, want_scores=True)
    assert nan_eq(got.v_score, ref["v"]) and nan_eq(got.f_score, ref["f"])
    assert got.ks.cpu().tolist() == ref["ks"].tolist()
    assert torch.equal(got.global_idx.cpu(), ref["global_idx"])


def test_video_centre_hand_over_is_race_free():
    """8 x 324 x 3584 fp16, seed 9428 (found by tests/tools/soak_gpu_vs_oracle.py): a video-centre mean sits EXACTLY on
    an fp16 rounding midpoint, so its replay decides a score -- and the replay's level-1 groups travel between
    workgroups of one launch.  A fence-free variant of that hand-over returned a stale group in one run of eight; the
    scores must equal the oracle's in every one of 40 runs."""
    x = make_input(8, 324, 3584, "f16", 9428, "drift")
    O.set_mode("torch")
    ref = O.compress_indices(x, 324, 0.25)
    xd = x.to(dev())
    for run in range(40):
        got = vc.compress(xd, 324, 0.25, want_scores=True)
        assert torch.equal(got.v_score.cpu(), ref["v"]) and torch.equal(got.f_score.cpu(), ref["f"]), f"run {run}"


@pytest.mark.parametrize("case", [(128, 196, 3584, "bf16", "drift"), (32, 196, 3584, "f16", "iid"), (64, 324, 1024, "bf16", "cancel")],
                         ids=lambda c: "x".join(map(str, c)))
def test_runs_are_bitwise_repeatable(case):
    """DESIGN.md "Determinism": no floating-point atomics, fixed reduction orders, the hand-overs between workgroups
    fenced -- 25 runs of one input (default mode, then the proven-margin mode with its many replays) give the same
    bytes: scores, budgets, kept indices, kept rows."""
    F, N, D, dn, dist = case
    xd = make_input(F, N, D, dn, 11, dist).to(dev())
    for mode in ("torch", "torch_proven"):
        try:
            _ffi.set_mode(mode)
            first = None
            for run in range(25 if mode == "torch" else 8):
                got = vc.compress(xd, N, 0.25, want_scores=True)
                sig = (synth.sha256_tensor(got.v_score.cpu()), synth.sha256_tensor(got.f_score.cpu()),
                       synth.sha256_tensor(got.global_idx.cpu()), synth.sha256_tensor(got.rows.cpu()), got.ks.cpu().tolist())
                if first is None:
                    first = sig
                assert sig == first, f"{mode}: run {run} differs from run 0"
        finally:
            _ffi.set_mode("torch")


@pytest.mark.gpu
def test_early_count_mirror_and_spare_outputs_change_nothing():
    """The one-shot API takes K from the pinned host mirror the selection launch writes (vc2_compress_ex2) and allocates
    the next call's outputs while the pass runs: same rows / indices / budgets as the blocking copy, call after call,
    and a result handed out earlier is never overwritten."""
    import vidcom2_amd as vc
    from vidcom2_amd import vidcom2 as V
    dev = torch.device("cuda:0")
    xs = [synth.make(16, 196, 1024, torch.bfloat16, sd, "drift").to(dev) for sd in (0, 1, 2)]
    old = (V._EARLY_COUNT, V._PREALLOC)
    try:
        V._EARLY_COUNT, V._PREALLOC = False, False
        V.clear_plan_cache()
        ref = [vc.compress(x, 196, 0.25) for x in xs]
        V._EARLY_COUNT, V._PREALLOC = True, True
        V.clear_plan_cache()
        got = [vc.compress(x, 196, 0.25) for x in xs for _ in range(2)][::2]
        torch.cuda.synchronize()
        for a, b in zip(ref, got):
            assert a.K == b.K and torch.equal(a.global_idx, b.global_idx) and torch.equal(a.ks, b.ks)
            assert torch.equal(a.rows, b.rows)
        assert len({g.rows.data_ptr() for g in got}) == len(got)
    finally:
        V._EARLY_COUNT, V._PREALLOC = old
        V.clear_plan_cache()


@pytest.mark.gpu
def test_a_guard_hit_of_any_frame_reaches_the_caller():
    """Status bit 4 (a loop bound of the selection replay expired) raised from the workgroup of a NON-last frame -- the
    last frame's workgroup writes the count and the early status words before the other selections end.  With the host
    mirror armed the hit arrives in the mirror's FINAL word (K_host[2]): taken by finish() if it is already there, else
    raised when the plan's next pass is enqueued (_settle).  Without the mirror the blocking copy sees it.  The stage
    call (vc2_select: no pass-wide status word) reports it in K_out[1] as well."""
    import vidcom2_amd as vc
    from vidcom2_amd import vidcom2 as V
    dev = torch.device("cuda:0")
    F, N, D = 16, 196, 1024
    x = synth.make(F, N, D, torch.bfloat16, 3, "drift").to(dev)
    L = _ffi.lib()
    try:
        for mirror in (False, True):
            plan = V.CompressPlan(F, N, D, torch.bfloat16, dev, 0.25)
            plan.enqueue(x, mirror=mirror); good = plan.finish(); plan._settle()
            L.vc2_selftest_force_guard(3)                      # frame 3 of 16: not the workgroup that writes the count
            plan.new_outputs(); plan.enqueue(x, mirror=mirror)
            raised = False
            try:
                plan.finish()
            except RuntimeError as e:
                raised = "selection replay" in str(e)
            if not raised:                                     # the early words were clean: the final word must not be
                assert mirror
                torch.cuda.synchronize()
                with pytest.raises(RuntimeError, match="PREVIOUS pass.*selection replay"):
                    plan.new_outputs(); plan.enqueue(x, mirror=mirror)
            L.vc2_selftest_force_guard(-1)
            torch.cuda.synchronize()
            plan._late = None
            plan.new_outputs(); plan.enqueue(x, mirror=mirror); again = plan.finish(); plan._settle()
            assert again.K == good.K and torch.equal(again.global_idx, good.global_idx)
        # the stage call: K_out[1] carries the bit whichever workgroup stored the count
        scores = torch.rand(F, N, device=dev).to(torch.bfloat16)
        scales = torch.full((F,), 0.25, device=dev).to(torch.bfloat16)
        L.vc2_selftest_force_guard(5)
        with pytest.raises(RuntimeError, match="selection replay"):
            vc.select_outlier_indices(scores, scales, N)
        L.vc2_selftest_force_guard(-1)
        assert [int(t.numel()) for t in vc.select_outlier_indices(scores, scales, N)] == [49] * F
    finally:
        L.vc2_selftest_force_guard(-1)
        c = (ctypes.c_int32 * 8)()
        L.vc2_selftest_counters(c, 1)


@pytest.mark.parametrize("shape", [(16, 196, 3584, "bf16", "drift"), (9, 100, 4096, "f16", "iid"), (12, 49, 1024, "bf16", "cancel"),
                                   (5, 324, 3584, "f16", "drift"), (3, 37, 200, "bf16", "iid"), (7, 64, 2048, "f16", "cancel")],
                         ids=lambda s_: "x".join(map(str, s_)))
def test_selection_engines_and_launch_forms_agree(shape):
    """Round 6 changed HOW the channel selection runs, not what it computes: thread-contiguous register rounds
    (k_chan_select4 / sel4_round; VC2_SEL4=0: the 16-wave LDS rounds of rounds 3-5; VC2_SEL3=1: the ballot form) and the
    variance reduction + selection in ONE launch with polled variances (k_var_select; VC2_VARSEL=0: two launches).  Every
    form must give the same channel order, scores, budgets and kept indices -- and the oracle's."""
    import os
    F, N, D, dn, dist = shape
    x = make_input(F, N, D, dn, 17, dist)
    O.set_mode("torch")
    ref = O.compress_indices(x, N, 0.25)
    xd = x.to(dev())
    keys = ("VC2_SEL4", "VC2_VARSEL", "VC2_SEL3")
    old = {k: os.environ.get(k) for k in keys}
    outs = []
    try:
        for form in ({}, {"VC2_VARSEL": "0"}, {"VC2_SEL4": "0"}, {"VC2_SEL3": "1"}):
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(form)
            r = vc.compress(xd, N, 0.25, want_scores=True)
            chan = vc.vidcom2.low_var_channel_order(xd)       # (the stage call: channel ORDER as torch.topk returns it)
            torch.cuda.synchronize()
            outs.append((form, r, chan))
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]
    for form, r, chan in outs:
        assert r.ks.cpu().tolist() == ref["ks"].tolist(), form
        assert torch.equal(r.global_idx.cpu(), ref["global_idx"]), form
        assert nan_eq(r.v_score, ref["v"]) and nan_eq(r.f_score, ref["f"]), form
        assert torch.equal(chan.cpu(), outs[0][2].cpu()), form


@pytest.mark.gpu
def test_a_late_guard_hit_is_reported_when_the_plan_is_dropped():
    """VERDICT r5 item 7 / ADVICE r5: a pass that returned on the early words of the host mirror and whose plan is never
    used again -- dropped, cleared out of (or evicted from) the plan cache, or followed by an un-mirrored enqueue -- still
    gets its final status word read: a RuntimeWarning where nothing may raise, the exception in front of ANY next pass."""
    import gc
    import warnings as W
    from vidcom2_amd import vidcom2 as V
    dev = torch.device("cuda:0")
    F, N, D = 16, 196, 1024
    x = synth.make(F, N, D, torch.bfloat16, 3, "drift").to(dev)
    L = _ffi.lib()

    def late_pass():
        """a plan whose last pass returned early with a guard hit still on its way (None: finish() already saw it)"""
        plan = V.CompressPlan(F, N, D, torch.bfloat16, dev, 0.25)
        plan.enqueue(x, mirror=True); plan.finish(); plan._settle()
        L.vc2_selftest_force_guard(3)
        plan.new_outputs(); plan.enqueue(x, mirror=True)
        try:
            plan.finish()
        except RuntimeError:
            plan = None
        L.vc2_selftest_force_guard(-1)
        torch.cuda.synchronize()
        return plan
    try:
        seen = 0
        for how in ("del", "quiet", "plain_enqueue") * 4:
            plan = late_pass()
            if plan is None or plan._late is None:
                continue
            seen += 1
            if how == "plain_enqueue":                           # an un-mirrored next pass settles the previous one too
                with pytest.raises(RuntimeError, match="PREVIOUS pass.*selection replay"):
                    plan.new_outputs(); plan.enqueue(x, mirror=False)
                plan._late = None
                continue
            with W.catch_warnings(record=True) as rec:
                W.simplefilter("always")
                if how == "del":
                    del plan; gc.collect()
                else:
                    plan.settle_quietly()
            assert any("selection replay" in str(r.message) for r in rec), how
        # (the early words normally precede the final one: at least one of the twelve tries must have been late)
        assert seen >= 1
        # the plan cache: clearing it settles its plans
        V.clear_plan_cache()
        good = vc.compress(x, N, 0.25)
        L.vc2_selftest_force_guard(3)
        try:
            vc.compress(x, N, 0.25)
            late = True
        except RuntimeError:
            late = False
        L.vc2_selftest_force_guard(-1)
        torch.cuda.synchronize()
        with W.catch_warnings(record=True) as rec:
            W.simplefilter("always")
            V.clear_plan_cache()
        assert (not late) or any("selection replay" in str(r.message) for r in rec)
        assert torch.equal(vc.compress(x, N, 0.25).global_idx, good.global_idx)
    finally:
        L.vc2_selftest_force_guard(-1)
        c = (ctypes.c_int32 * 8)()
        L.vc2_selftest_counters(c, 1)
        V.clear_plan_cache()


@pytest.mark.gpu
def test_plan_cache_is_safe_under_threads():
    """Several threads through the one-shot API at once, more shapes than the cache holds (evictions while others look
    up): no KeyError, every result equal to the single-threaded one."""
    import threading
    import vidcom2_amd as vc
    from vidcom2_amd import vidcom2 as V
    dev = torch.device("cuda:0")
    shapes = [(4, 49, 64), (6, 49, 128), (8, 16, 256), (5, 100, 64), (3, 196, 128), (7, 37, 64)]
    xs = [synth.make(F, N, D, torch.bfloat16, 7, "drift").to(dev) for F, N, D in shapes]
    want = [vc.compress(x, s[1], 0.25).global_idx.cpu() for x, s in zip(xs, shapes)]
    old_max = V._PLAN_CACHE_MAX
    errs = []

    def work(t):
        try:
            for it in range(30):
                i = (t + it) % len(shapes)
                r = vc.compress(xs[i], shapes[i][1], 0.25)
                if not torch.equal(r.global_idx.cpu(), want[i]):
                    errs.append(("mismatch", t, i))
        except Exception as e:  # noqa: BLE001
            errs.append((repr(e), t))
    try:
        V._PLAN_CACHE_MAX = 3
        V.clear_plan_cache()
        th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
    finally:
        V._PLAN_CACHE_MAX = old_max
        V.clear_plan_cache()
    assert not errs, errs[:3]
