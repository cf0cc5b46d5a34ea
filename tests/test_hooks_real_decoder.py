"""The Qwen2.5-VL hook through the installed transformers' REAL decoder stack (tiny random-init model on CPU; the
device pass is replaced by the oracle): the hooked forward must equal running the decoder by hand on the pruned
prompt with positions computed BEFORE pruning (the reference's semantics, models/qwen2_5_vl.py:89-183)."""
import os
import types

import pytest
import torch

import _stub_models as S
import oracle as O

D = 64


def _model():
    from transformers.models.qwen2_5_vl import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLModel
    vision = dict(depth=1, hidden_size=32, intermediate_size=32, num_heads=2, out_hidden_size=D,
                  fullatt_block_indexes=[0], spatial_merge_size=2)
    text = dict(vocab_size=S.VOCAB, hidden_size=D, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=2, max_position_embeddings=4096, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                rope_parameters=dict(rope_type="default", mrope_section=[4, 6, 6], rope_theta=10000.0))
    cfg = Qwen2_5_VLConfig(text_config=text, vision_config=vision, image_token_id=S.IMAGE_ID, video_token_id=S.VIDEO_ID)
    torch.manual_seed(0)
    return Qwen2_5_VLModel(cfg).eval()


@pytest.fixture()
def hooked(monkeypatch):
    import vidcom2_amd.models.qwen2_5_vl as H

    def keep(feat, g, m, base):
        t, h, w = g.tolist()
        return O.compress_indices(feat.contiguous(), (h * w) // (m * m), base)["global_idx"]

    O.set_mode("torch")
    monkeypatch.setattr(H, "_compute_keep_indices", keep)
    monkeypatch.setenv("COMPRESSOR", "vidcom2")
    monkeypatch.setenv("R_RATIO", "0.5")
    return H.Qwen2_5_VLModel_forward


@pytest.mark.parametrize("with_types", [True, False], ids=["mrope_positions", "decoder_inferred_positions"])
@torch.no_grad()
def test_hook_equals_manual_pruned_run(hooked, with_types):
    c = dict(D=D, dt="f32", seed=22, r="0.5", grids=[[4, 8, 8], [6, 12, 8]], prefix=7, between=3, suffix=5, mask="2d")
    ids, feats, _, mask = S.qwen_inputs(c)
    grid = torch.tensor(c["grids"])
    model = _model()
    S.set_video_features(model, feats)
    mm = (ids == S.VIDEO_ID).int() * 2 if with_types else None
    kw = dict(input_ids=ids, attention_mask=mask, pixel_values_videos=torch.zeros(1, 4), video_grid_thw=grid,
              mm_token_type_ids=mm)
    dense = model(**kw).last_hidden_state
    model.forward = types.MethodType(hooked, model)
    out = model(**kw).last_hidden_state
    st = model._vidcom2_last
    keep = st.keep_token_indices
    assert st.pruned and out.shape[1] == keep.numel() < dense.shape[1]

    emb = model.get_input_embeddings()(ids).clone()
    emb[ids == S.VIDEO_ID] = torch.cat(feats)
    if with_types:
        model.rope_deltas = None
        pos = model.compute_3d_position_ids(input_ids=ids, image_grid_thw=None, video_grid_thw=grid, inputs_embeds=emb,
                                            attention_mask=mask, past_key_values=None, mm_token_type_ids=mm)
        pos = pos[..., keep]
    else:
        pos = torch.cat((torch.arange(keep.numel()).view(1, 1, -1), keep.view(1, 1, -1).expand(3, 1, -1)), dim=0)
    manual = model.language_model(inputs_embeds=emb[:, keep], position_ids=pos, attention_mask=mask[:, keep])
    assert torch.allclose(out, manual.last_hidden_state, atol=1e-6)
    # and pruning really changes what the decoder sees
    assert not torch.allclose(out, dense[:, keep], atol=1e-3)


@torch.no_grad()
def test_generate_runs_with_pruned_prefill(hooked):
    """`generate()` of the installed transformers with the hook bound on the inner model: the prefill is pruned, the
    decode steps run on the shorter KV cache, and the first new token is the one the manual pruned prefill predicts."""
    from transformers.models.qwen2_5_vl import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLForConditionalGeneration
    from vidcom2_amd.models import install
    inner_cfg = _model().config
    torch.manual_seed(0)
    model = Qwen2_5_VLForConditionalGeneration(inner_cfg).eval()
    c = dict(D=D, dt="f32", seed=22, r="0.5", grids=[[4, 8, 8], [6, 12, 8]], prefix=7, between=3, suffix=5, mask="2d")
    ids, feats, _, mask = S.qwen_inputs(c)
    grid = torch.tensor(c["grids"])
    S.set_video_features(model.model, feats)
    mm = (ids == S.VIDEO_ID).int() * 2
    kw = dict(input_ids=ids, attention_mask=mask, pixel_values_videos=torch.zeros(1, 4), video_grid_thw=grid,
              mm_token_type_ids=mm, max_new_tokens=3, do_sample=False, eos_token_id=None, pad_token_id=0)
    assert install(model)
    out = model.generate(**kw)
    assert out.shape == (1, ids.shape[1] + 3)
    st = model.model._vidcom2_last
    assert st.pruned
    keep = st.keep_token_indices
    emb = model.model.get_input_embeddings()(ids).clone()
    emb[ids == S.VIDEO_ID] = torch.cat(feats)
    model.model.rope_deltas = None
    pos = model.model.compute_3d_position_ids(input_ids=ids, image_grid_thw=None, video_grid_thw=grid, inputs_embeds=emb,
                                              attention_mask=mask, past_key_values=None, mm_token_type_ids=mm)
    h = model.model.language_model(inputs_embeds=emb[:, keep], position_ids=pos[..., keep],
                                   attention_mask=mask[:, keep]).last_hidden_state
    first = model.lm_head(h[:, -1]).argmax(-1)
    assert int(first) == int(out[0, ids.shape[1]])


@torch.no_grad()
def test_qwen3_hook_through_real_decoder_with_deepstack(hooked):
    from transformers.models.qwen3_vl import Qwen3VLConfig
    from transformers.models.qwen3_vl.modeling_qwen3_vl import Qwen3VLModel
    from vidcom2_amd.models.qwen3_vl import Qwen3VLModel_forward
    vision = dict(depth=1, hidden_size=32, intermediate_size=32, num_heads=2, out_hidden_size=D, spatial_merge_size=2,
                  deepstack_visual_indexes=[0], num_position_embeddings=16)
    text = dict(vocab_size=S.VOCAB, hidden_size=D, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=2, head_dim=32, max_position_embeddings=4096, bos_token_id=1, eos_token_id=2,
                pad_token_id=0, rope_parameters=dict(rope_type="default", mrope_section=[4, 6, 6], rope_theta=10000.0,
                                                     mrope_interleaved=True))
    cfg = Qwen3VLConfig(text_config=text, vision_config=vision, image_token_id=S.IMAGE_ID, video_token_id=S.VIDEO_ID)
    torch.manual_seed(0)
    model = Qwen3VLModel(cfg).eval()
    c = dict(D=D, dt="f32", seed=22, r="0.5", grids=[[4, 8, 8], [6, 12, 8]], prefix=7, between=3, suffix=5, mask="2d")
    ids, feats, pos, mask = S.qwen_inputs(c)
    deep = S.deepstack_feats(sum(f.shape[0] for f in feats), D, torch.float32, 5, layers=1)
    S.set_video_features(model, feats, deep)
    kw = dict(input_ids=ids, attention_mask=mask, position_ids=pos, pixel_values_videos=torch.zeros(1, 4),
              video_grid_thw=torch.tensor(c["grids"]))
    model.forward = types.MethodType(Qwen3VLModel_forward, model)
    out = model(**kw).last_hidden_state
    st = model._vidcom2_last
    keep = st.keep_token_indices
    assert st.pruned and out.shape[1] == keep.numel()
    emb = model.get_input_embeddings()(ids).clone()
    vm = ids == S.VIDEO_ID
    emb[vm] = torch.cat(feats)
    flags = torch.zeros(ids.shape[1], dtype=torch.bool)
    flags[keep] = True
    manual = model.language_model(inputs_embeds=emb[:, keep], position_ids=pos[..., keep], attention_mask=mask[:, keep],
                                  visual_pos_masks=vm[:, keep], deepstack_visual_embeds=[d[flags[vm[0]]] for d in deep])
    assert torch.allclose(out, manual.last_hidden_state, atol=1e-6)


@torch.no_grad()
def test_qwen2vl_hook_through_real_decoder(monkeypatch):
    from transformers.models.qwen2_vl import Qwen2VLConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLModel
    import vidcom2_amd.models.qwen2_vl as H2
    O.set_mode("torch")
    monkeypatch.setenv("R_RATIO", "0.5")
    monkeypatch.setattr(H2, "_keep_index", lambda merged, g, ms=2: O.compress_indices(
        merged.contiguous(), int((g[:, 1] // ms) * (g[:, 2] // ms)), 0.5)["global_idx"])
    vision = dict(depth=1, embed_dim=32, hidden_size=D, num_heads=2, mlp_ratio=1, spatial_merge_size=2)
    text = dict(vocab_size=S.VOCAB, hidden_size=D, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=2, max_position_embeddings=4096, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                rope_parameters=dict(rope_type="default", mrope_section=[4, 6, 6], rope_theta=10000.0))
    cfg = Qwen2VLConfig(text_config=text, vision_config=vision, image_token_id=S.IMAGE_ID, video_token_id=S.VIDEO_ID)
    torch.manual_seed(0)
    model = Qwen2VLModel(cfg).eval()
    c = dict(D=D, dt="f32", seed=23, r="0.5", grids=[[6, 12, 8]], prefix=7, between=0, suffix=5, mask="2d")
    ids, feats, _, mask = S.qwen_inputs(c)
    S.set_video_features(model, feats)
    model.forward = types.MethodType(H2.Qwen2VLModel_forward, model)
    out = model(input_ids=ids, attention_mask=mask, pixel_values_videos=torch.zeros(1, 4),
                video_grid_thw=torch.tensor(c["grids"]), mm_token_type_ids=(ids == S.VIDEO_ID).int() * 2)
    st = model._vidcom2_last
    assert st.pruned and out.last_hidden_state.shape[1] == st.keep_token_indices.numel() < ids.shape[1]
