"""Model hooks (vidcom2_amd/models) against the reference hooks' recorded behaviour.

Fixtures: tests/golden/hook_cases.json (made by make_hook_golden.py from the reference's own
`cus_prepare_inputs_labels_for_multimodal` / `Qwen2_5_VLModel_forward`).

* `-m "not gpu"`: the hook *logic* (interception, gating, pruning, restoration); the device pass is
  replaced by the CPU oracle via monkeypatch -- the one place outside the product where that is
  allowed, it is the checker here.
* `-m gpu`: the same cases end to end through the HIP path.
"""
import os
import types

import pytest
import torch

import _stub_models as S
import oracle as O
from conftest import load_json
from vidcom2_amd import synth

CASES = load_json("hook_cases.json")
sha = synth.sha256_tensor


@pytest.fixture(autouse=True)
def _env():
    old = {k: os.environ.get(k) for k in ("COMPRESSOR", "R_RATIO")}
    O.set_mode("torch")
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _use_oracle(monkeypatch):
    """Swap the device pass under the hooks for the oracle (CPU tests only)."""
    import vidcom2_amd.models.llava as HL
    import vidcom2_amd.models.qwen2_5_vl as HQ
    import vidcom2_amd.models.qwen2_vl as HQ2

    def keep(feat, grid, merge, base):
        t, h, w = (int(v) for v in grid.tolist())
        n = (h * w) // (merge ** 2)
        if n <= 0 or feat.numel() == 0:
            return torch.arange(feat.shape[0])
        return O.compress_indices(feat.contiguous(), n, base)["global_idx"]

    def keep_q2(merged, grid_thw, merge_size=2):
        tpf = int((grid_thw[:, 1] // merge_size) * (grid_thw[:, 2] // merge_size))
        return O.compress_indices(merged.contiguous(), tpf, float(os.getenv("R_RATIO", "0.25")))["global_idx"]

    monkeypatch.setattr(HQ, "_compute_keep_indices", keep)
    monkeypatch.setattr(HQ2, "_keep_index", keep_q2)
    monkeypatch.setattr(HL, "vidcom2_compression",
                        lambda flat, model="llava_ov", base_scale=0.25, frame_token_len=None, img_feat=None:
                        O.vidcom2_compression(flat.contiguous(), model, base_scale, frame_token_len, img_feat))


# ------------------------------------------------------------------------------------------
# LLaVA
# ------------------------------------------------------------------------------------------
def _run_llava(c, device):
    from vidcom2_amd.models.llava import cus_prepare_inputs_labels_for_multimodal as hook
    os.environ["R_RATIO"] = c["r"]
    ids, feats = S.llava_inputs(c)
    ids, feats = ids.to(device), feats.to(device)
    m = S.StubLlava(c["D"], S.DT[c["dt"]], device, c["seed"], c["merge"], c["newline"])
    plain = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)
    m.prepare_inputs_labels_for_multimodal = types.MethodType(hook, m)
    comp = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)
    # nothing left behind on the instance
    assert "add_token_per_grid" not in m.__dict__ and "get_2dPool" not in m.__dict__
    return plain, comp


def _check_llava(c, plain, comp):
    assert list(plain[4].shape) == c["plain_shape"] and sha(plain[4]) == c["plain_sha"]
    assert list(comp[4].shape) == c["comp_shape"]
    assert sha(comp[4]) == c["comp_sha"]
    assert comp[0] is None and comp[1] is None and comp[2] is None and comp[5] is None
    assert type(comp[4]) is torch.Tensor


@pytest.mark.parametrize("c", CASES["llava"], ids=lambda c: c["name"])
def test_llava_hook_logic_cpu(c, monkeypatch):
    _use_oracle(monkeypatch)
    _check_llava(c, *_run_llava(c, "cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES["llava"], ids=lambda c: c["name"])
def test_llava_hook_gpu(c):
    _check_llava(c, *_run_llava(c, "cuda"))


def test_llava_hook_restores_on_error(monkeypatch):
    from vidcom2_amd.models import llava as HL
    monkeypatch.setattr(HL, "vidcom2_compression", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("boom")))
    c = CASES["llava"][0]
    ids, feats = S.llava_inputs(c)
    m = S.StubLlava(c["D"], S.DT[c["dt"]], "cpu", c["seed"], c["merge"], c["newline"])
    m.prepare_inputs_labels_for_multimodal = types.MethodType(HL.cus_prepare_inputs_labels_for_multimodal, m)
    with pytest.raises(RuntimeError, match="boom"):
        m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)
    assert "add_token_per_grid" not in m.__dict__


def test_llava_hook_decode_step_and_no_images():
    from vidcom2_amd.models.llava import cus_prepare_inputs_labels_for_multimodal as hook
    m = S.StubLlava(64, torch.float32, "cpu", 1)
    m.prepare_inputs_labels_for_multimodal = types.MethodType(hook, m)
    ids = torch.tensor([[5]])
    out = m.prepare_inputs_labels_for_multimodal(ids, None, None, "kv", None, None)
    assert out[0] is ids and out[3] == "kv" and out[4] is None


def test_llava_hook_wrong_tokens_per_frame_raises_like_reference(monkeypatch):
    """one_token path with N != 196: the reference dies in x.view(-1, 196, C) (vidcom2.py:47)."""
    _use_oracle(monkeypatch)
    from vidcom2_amd.models.llava import cus_prepare_inputs_labels_for_multimodal as hook
    m = S.StubLlava(64, torch.float32, "cpu", 1, "spatial_unpad", "one_token")
    m.prepare_inputs_labels_for_multimodal = types.MethodType(hook, m)
    feats = S.video_feats(3, 100, 64, torch.float32, 2)[..., None]
    ids = torch.tensor([[1, S.IMAGE_TOKEN_INDEX, 2]])
    with pytest.raises(RuntimeError):
        m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)


def test_hook_needs_an_original():
    from vidcom2_amd.models._intercept import original_method
    from vidcom2_amd.models.llava import cus_prepare_inputs_labels_for_multimodal as hook

    class Bare:
        pass

    with pytest.raises(AttributeError, match="no original"):
        original_method(Bare(), "prepare_inputs_labels_for_multimodal", hook)

    class Sub(S.StubLlava):                    # hook assigned on a subclass: the parent's method is the original
        prepare_inputs_labels_for_multimodal = hook

    fn = original_method(Sub(64, torch.float32, "cpu", 1), "prepare_inputs_labels_for_multimodal", hook)
    assert fn.__func__ is S.StubLlava.prepare_inputs_labels_for_multimodal


# ------------------------------------------------------------------------------------------
# Qwen2.5-VL (the installed transformers' real Qwen2_5_VLModel)
# ------------------------------------------------------------------------------------------
def _run_qwen(family, hook, c, device, compressor="vidcom2", **extra):
    os.environ["R_RATIO"] = c["r"]
    if compressor is None:
        os.environ.pop("COMPRESSOR", None)
    else:
        os.environ["COMPRESSOR"] = compressor
    dtype = S.DT[c["dt"]]
    ids, feats, pos, mask = S.qwen_inputs(c)
    model, rec = S.make_qwen_vl_model(family, c["D"], dtype, device, c["seed"])
    S.set_video_features(model, [f.to(device) for f in feats])
    model.forward = types.MethodType(hook, model)
    kwargs = dict(input_ids=ids.to(device), attention_mask=None if mask is None else mask.to(device),
                  position_ids=pos.to(device), pixel_values_videos=torch.zeros(1, 4, device=device),
                  video_grid_thw=torch.tensor(c["grids"], device=device))
    kwargs.update(extra)
    out = model(**kwargs)
    assert "language_model" not in model.__dict__ and "get_placeholder_mask" not in model.__dict__
    return model, rec.calls[-1], out


def _check_qwen25(c, seen):
    assert seen["position_ids"][0, 0].tolist() == c["keep_token_indices"]
    assert torch.equal(seen["position_ids"][1], seen["position_ids"][0])
    assert sha(seen["inputs_embeds"]) == c["embeds_sha"]
    if c["mask_sha"] is None:
        assert seen["attention_mask"] is None
    else:
        assert list(seen["attention_mask"].shape) == c["mask_shape"]
        assert sha(seen["attention_mask"]) == c["mask_sha"]
    assert seen["input_ids"] is None


@pytest.mark.parametrize("c", CASES["qwen2_5_vl"], ids=lambda c: c["name"])
def test_qwen25_hook_logic_cpu(c, monkeypatch):
    _use_oracle(monkeypatch)
    from vidcom2_amd.models.qwen2_5_vl import Qwen2_5_VLModel_forward as hook
    model, seen, out = _run_qwen("qwen2_5_vl", hook, c, "cpu")
    _check_qwen25(c, seen)
    assert model._vidcom2_last.pruned
    assert out.last_hidden_state.shape[1] == len(c["keep_token_indices"])


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES["qwen2_5_vl"], ids=lambda c: c["name"])
def test_qwen25_hook_gpu(c):
    from vidcom2_amd.models.qwen2_5_vl import Qwen2_5_VLModel_forward as hook
    _, seen, _ = _run_qwen("qwen2_5_vl", hook, c, "cuda")
    _check_qwen25(c, seen)


class _FullCache:
    def get_seq_length(self):
        return 7


@pytest.mark.parametrize("why", ["env_off", "decode_step", "no_video"])
def test_qwen25_hook_gating(why, monkeypatch):
    """qwen2_5_vl.py:120-129 -- anything but a batch-1 video prefill with COMPRESSOR=vidcom2 runs unpruned."""
    _use_oracle(monkeypatch)
    from vidcom2_amd.models.qwen2_5_vl import Qwen2_5_VLModel_forward as hook
    c = CASES["qwen2_5_vl"][0]
    if why == "env_off":
        _, seen, _ = _run_qwen("qwen2_5_vl", hook, c, "cpu", compressor=None)
    elif why == "decode_step":
        _, seen, _ = _run_qwen("qwen2_5_vl", hook, c, "cpu", past_key_values=_FullCache())
    else:
        _, seen, _ = _run_qwen("qwen2_5_vl", hook, c, "cpu", pixel_values_videos=None)
    assert seen["inputs_embeds"].shape[1] == c["seq_len"]
    assert seen["position_ids"].shape[-1] == c["seq_len"]


def test_qwen25_hook_batch_of_two_is_left_alone(monkeypatch):
    _use_oracle(monkeypatch)
    from vidcom2_amd.models.qwen2_5_vl import Qwen2_5_VLModel_forward as hook
    c = CASES["qwen2_5_vl"][0]
    os.environ["R_RATIO"], os.environ["COMPRESSOR"] = c["r"], "vidcom2"
    ids, feats, pos, mask = S.qwen_inputs(c)
    model, rec = S.make_qwen_vl_model("qwen2_5_vl", c["D"], S.DT[c["dt"]], "cpu", c["seed"])
    S.set_video_features(model, [feats[0], feats[0]])
    model.forward = types.MethodType(hook, model)
    model(input_ids=ids.repeat(2, 1), attention_mask=mask.repeat(2, 1), position_ids=pos.repeat(1, 2, 1),
          pixel_values_videos=torch.zeros(1, 4), video_grid_thw=torch.tensor(c["grids"] * 2))
    assert rec.calls[-1]["inputs_embeds"].shape[:2] == (2, c["seq_len"])
    assert not model._vidcom2_last.pruned


def test_compute_keep_indices_degenerate():
    """qwen2_5_vl.py:26-28: empty clip / zero tokens per frame -> keep everything (no device needed)."""
    from vidcom2_amd.models.qwen2_5_vl import _compute_keep_indices
    x = torch.zeros(0, 64)
    assert _compute_keep_indices(x, torch.tensor([2, 2, 2]), 2, 0.25).numel() == 0
    x = torch.zeros(6, 64)
    assert _compute_keep_indices(x, torch.tensor([2, 1, 1]), 2, 0.25).tolist() == list(range(6))


@pytest.mark.gpu
def test_compute_keep_indices_matches_stage_chain():
    """The fused pass under the hook == the reference's chain of stage calls (qwen2_5_vl.py:29-33)."""
    import vidcom2_amd as V
    from vidcom2_amd.models.qwen2_5_vl import _compute_keep_indices
    x = synth.make(12, 36, 256, torch.bfloat16, 5).cuda()
    grid = torch.tensor([12, 12, 12])
    got = _compute_keep_indices(x, grid, 2, 0.3)
    sel = V.select_low_var_channels(x)
    v, f = V.compute_gaussian_scores(sel, 36)
    scales = V.compute_scales(-v.mean(dim=-1), 0.3)
    want = V._map_linear_offset(V.select_outlier_indices(v + f, scales, 36), 36)
    assert torch.equal(got, want)
    ref = O.compress_indices(x.cpu(), 36, 0.3)["global_idx"]
    assert torch.equal(got.cpu(), ref)


# ------------------------------------------------------------------------------------------
# Qwen2-VL (no reference fixture: the reference hook is stale, see models/qwen2_vl.py)
# ------------------------------------------------------------------------------------------
def _expect_q2(c):
    ids, feats, pos, mask = S.qwen_inputs(c)
    t, h, w = c["grids"][0]
    kept = O.compress_indices(feats[0], (h // 2) * (w // 2), float(c["r"]))["global_idx"]
    vm = ids[0] == S.VIDEO_ID
    first = int(vm.nonzero()[0])
    keep = torch.cat((torch.arange(first), kept + first, torch.arange(first + feats[0].shape[0], ids.shape[1])))
    return keep


Q2_CASES = [c for c in CASES["qwen2_5_vl"] if len(c["grids"]) == 1]


@pytest.mark.parametrize("c", Q2_CASES, ids=lambda c: c["name"])
def test_qwen2vl_model_hook_logic_cpu(c, monkeypatch):
    _use_oracle(monkeypatch)
    from vidcom2_amd.models.qwen2_vl import Qwen2VLModel_forward as hook
    _, seen, _ = _run_qwen("qwen2_vl", hook, c, "cpu", compressor=None)      # always on, like the reference
    assert seen["position_ids"][0, 0].tolist() == _expect_q2(c).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("c", Q2_CASES, ids=lambda c: c["name"])
def test_qwen2vl_model_hook_gpu(c):
    from vidcom2_amd.models.qwen2_vl import Qwen2VLModel_forward as hook
    _, seen, _ = _run_qwen("qwen2_vl", hook, c, "cuda", compressor=None)
    assert seen["position_ids"][0, 0].tolist() == _expect_q2(c).tolist()


def test_qwen2vl_two_videos_is_an_error_like_the_reference(monkeypatch):
    _use_oracle(monkeypatch)
    from vidcom2_amd.models.qwen2_vl import Qwen2VLModel_forward as hook
    c = CASES["qwen2_5_vl"][1]
    with pytest.raises((ValueError, RuntimeError)):
        _run_qwen("qwen2_vl", hook, c, "cpu")


class _LegacyTower:
    """transformers-4.4x style vision tower: forward(hidden_states, grid_thw) -> merged embeddings."""
    spatial_merge_size = 2

    def __init__(self, feats):
        self.feats = feats

    def forward(self, hidden_states, grid_thw):
        return self.feats

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def get_dtype(self):
        return self.feats.dtype


class _LegacyGeneration:
    """transformers-4.4x style Qwen2VLForConditionalGeneration, reduced to what the hook brackets."""

    def __init__(self, c, tower, table):
        self.config = types.SimpleNamespace(video_token_id=S.VIDEO_ID, image_token_id=S.IMAGE_ID)
        self.visual, self.table, self.seen = tower, table, {}
        self.model = self._decoder

    def _decoder(self, **kw):
        self.seen = kw
        return (kw["inputs_embeds"],)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, pixel_values_videos=None,
                video_grid_thw=None):
        embeds = self.table[input_ids]
        if pixel_values_videos is not None:
            video = self.visual(pixel_values_videos, grid_thw=video_grid_thw)
            mask = (input_ids == self.config.video_token_id).unsqueeze(-1).expand_as(embeds)
            embeds = embeds.masked_scatter(mask, video.to(embeds.dtype))
        return self.model(input_ids=None, position_ids=position_ids, attention_mask=attention_mask,
                          inputs_embeds=embeds)


@pytest.mark.parametrize("c", Q2_CASES[:2], ids=lambda c: c["name"])
def test_qwen2vl_legacy_pair_logic_cpu(c, monkeypatch):
    """`Qwen2VL_ViT_forward` + `Qwen2VLGeneration_forward` installed the way the reference's are."""
    _use_oracle(monkeypatch)
    from vidcom2_amd.models.qwen2_vl import Qwen2VL_ViT_forward, Qwen2VLGeneration_forward
    os.environ["R_RATIO"] = c["r"]
    ids, feats, pos, mask = S.qwen_inputs(c)
    tower = _LegacyTower(feats[0])
    tower.forward = types.MethodType(Qwen2VL_ViT_forward, tower)
    merged, keep_index = tower(torch.zeros(1, 4), grid_thw=torch.tensor(c["grids"]))
    assert merged is feats[0] and keep_index.dtype == torch.int64
    gen = _LegacyGeneration(c, tower, S.embed_table(c["D"], S.DT[c["dt"]], c["seed"]))
    gen.forward = types.MethodType(Qwen2VLGeneration_forward, gen)
    gen.forward(input_ids=ids, attention_mask=mask, position_ids=pos, pixel_values_videos=torch.zeros(1, 4),
                video_grid_thw=torch.tensor(c["grids"]))
    want = _expect_q2(c)
    assert gen.seen["position_ids"][0, 0].tolist() == want.tolist()
    assert gen.seen["inputs_embeds"].shape[1] == want.numel()
    assert "visual" not in gen.__dict__ or gen.__dict__["visual"] is tower


# ------------------------------------------------------------------------------------------
# installer
# ------------------------------------------------------------------------------------------
def test_install_binds_by_family(monkeypatch):
    from vidcom2_amd.models import install
    from vidcom2_amd.models.llava import cus_prepare_inputs_labels_for_multimodal
    from vidcom2_amd.models.qwen2_5_vl import Qwen2_5_VLModel_forward
    from vidcom2_amd.models.qwen2_vl import Qwen2VLModel_forward
    os.environ.pop("COMPRESSOR", None)
    m = S.StubLlava(64, torch.float32, "cpu", 1)
    assert install(m) is False and "prepare_inputs_labels_for_multimodal" not in m.__dict__
    os.environ["COMPRESSOR"] = "vidcom2"
    assert install(m) and m.prepare_inputs_labels_for_multimodal.__func__ is cus_prepare_inputs_labels_for_multimodal
    q25, _ = S.make_qwen_vl_model("qwen2_5_vl", 64, torch.float32, "cpu", 1)
    assert install(q25) and q25.forward.__func__ is Qwen2_5_VLModel_forward
    q2, _ = S.make_qwen_vl_model("qwen2_vl", 64, torch.float32, "cpu", 1)
    assert install(q2) and q2.forward.__func__ is Qwen2VLModel_forward
    from vidcom2_amd.models.qwen3_vl import Qwen3VLModel_forward
    q3, _ = S.make_qwen_vl_model("qwen3_vl", 64, torch.float32, "cpu", 1)
    assert install(q3) and q3.forward.__func__ is Qwen3VLModel_forward
    wrapper = type("FakeForConditionalGeneration", (), {})()
    wrapper.model = S.make_qwen_vl_model("qwen2_5_vl", 64, torch.float32, "cpu", 1)[0]
    assert install(wrapper) and wrapper.model.forward.__func__ is Qwen2_5_VLModel_forward
    with pytest.raises(TypeError):
        install(object.__new__(type("Other", (), {})))


# ------------------------------------------------------------------------------------------
# Qwen3-VL (installed transformers' real Qwen3VLModel; deepstack features pruned with the prompt)
# ------------------------------------------------------------------------------------------
def _run_qwen3(c, device):
    from vidcom2_amd.models.qwen3_vl import Qwen3VLModel_forward as hook
    os.environ["R_RATIO"], os.environ["COMPRESSOR"] = c["r"], "vidcom2"
    dtype = S.DT[c["dt"]]
    ids, feats, pos, mask = S.qwen_inputs(c)
    n_vid = sum(f.shape[0] for f in feats)
    model, rec = S.make_qwen_vl_model("qwen3_vl", c["D"], dtype, device, c["seed"])
    S.set_video_features(model, [f.to(device) for f in feats],
                         [d.to(device) for d in S.deepstack_feats(n_vid, c["D"], dtype, c["seed"])])
    kwargs = dict(input_ids=ids.to(device), attention_mask=None if mask is None else mask.to(device),
                  position_ids=pos.to(device), pixel_values_videos=torch.zeros(1, 4, device=device),
                  video_grid_thw=torch.tensor(c["grids"], device=device))
    if c["n_image"]:
        img = S.video_feats(1, c["n_image"], c["D"], dtype, c["seed"] + 9)[0].to(device)
        S.set_image_features(model, [img], [d.to(device) for d in
                                            S.deepstack_feats(c["n_image"], c["D"], dtype, c["seed"] + 9)])
        kwargs.update(pixel_values=torch.zeros(1, 4, device=device),
                      image_grid_thw=torch.tensor([[1, 6, 6]], device=device))
    model.forward = types.MethodType(hook, model)
    model(**kwargs)
    return rec.calls[-1]


def _check_qwen3(c, seen):
    assert seen["position_ids"][0, 0].tolist() == c["keep_token_indices"]
    assert sha(seen["inputs_embeds"]) == c["embeds_sha"]
    if c["mask_sha"] is not None:
        assert sha(seen["attention_mask"]) == c["mask_sha"]
    assert sha(seen["visual_pos_masks"].to(torch.uint8)) == c["vpm_sha"]
    assert [list(d.shape) for d in seen["deepstack_visual_embeds"]] == c["deep_shapes"]
    assert [sha(d) for d in seen["deepstack_visual_embeds"]] == c["deep_sha"]
    assert int(seen["visual_pos_masks"].sum()) == c["deep_shapes"][0][0]


@pytest.mark.parametrize("c", CASES["qwen3_vl"], ids=lambda c: c["name"])
def test_qwen3_hook_logic_cpu(c, monkeypatch):
    _use_oracle(monkeypatch)
    _check_qwen3(c, _run_qwen3(c, "cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES["qwen3_vl"], ids=lambda c: c["name"])
def test_qwen3_hook_gpu(c):
    _check_qwen3(c, _run_qwen3(c, "cuda"))


def test_qwen_hook_four_row_position_ids_and_cache_position(monkeypatch):
    """transformers >= 4.5x layout: row 0 of a 4-row position tensor carries text positions for packed-sequence
    detection; after pruning it must stay contiguous, rows 1..3 (rope) are sliced; a prefill `cache_position` of the
    full prompt length shrinks with the prompt."""
    _use_oracle(monkeypatch)
    from vidcom2_amd.models.qwen2_5_vl import Qwen2_5_VLModel_forward as hook
    c = CASES["qwen2_5_vl"][0]
    ids, feats, pos3, mask = S.qwen_inputs(c)
    L = ids.shape[1]
    pos4 = torch.cat((torch.arange(L).view(1, 1, L), pos3 * 2 + 1), dim=0)
    _, seen, _ = _run_qwen("qwen2_5_vl", hook, c, "cpu", position_ids=pos4, cache_position=torch.arange(L))
    keep = torch.tensor(c["keep_token_indices"])
    got = seen["position_ids"]
    assert got.shape == (4, 1, keep.numel())
    assert got[0, 0].tolist() == list(range(keep.numel()))
    assert torch.equal(got[1:, 0], (pos3 * 2 + 1)[:, 0][:, keep])
    assert seen["cache_position"].tolist() == list(range(keep.numel()))


def test_hook_signatures_are_the_references():
    """inspect.signature(model.model.forward) on a hooked model shows the reference's explicit parameter lists
    (token_compressor/vidcom2/models/qwen2_5_vl.py:36-55, qwen3_vl.py:36-48, qwen2_vl.py:46-64 without `labels`), not
    (*args, **kwargs): HF generate's argument validation and tracing tools read them."""
    import inspect
    from vidcom2_amd.models import qwen2_5_vl, qwen2_vl, qwen3_vl
    q25 = ["self", "input_ids", "attention_mask", "position_ids", "past_key_values", "inputs_embeds", "use_cache",
           "output_attentions", "output_hidden_states", "return_dict", "pixel_values", "pixel_values_videos",
           "image_grid_thw", "video_grid_thw", "rope_deltas", "cache_position", "second_per_grid_ts", "kwargs"]
    q3 = ["self", "input_ids", "attention_mask", "position_ids", "past_key_values", "inputs_embeds", "pixel_values",
          "pixel_values_videos", "image_grid_thw", "video_grid_thw", "cache_position", "kwargs"]
    assert list(inspect.signature(qwen2_5_vl.Qwen2_5_VLModel_forward).parameters) == q25
    assert list(inspect.signature(qwen3_vl.Qwen3VLModel_forward).parameters) == q3
    assert list(inspect.signature(qwen2_vl.Qwen2VLModel_forward).parameters) == q25[:16] + ["kwargs"]
    for fn in (qwen2_5_vl.Qwen2_5_VLModel_forward, qwen3_vl.Qwen3VLModel_forward, qwen2_vl.Qwen2VLModel_forward):
        ps = inspect.signature(fn).parameters
        assert all(p.default is None for n, p in ps.items() if n not in ("self", "kwargs"))
        assert ps["kwargs"].kind is inspect.Parameter.VAR_KEYWORD
