#!/usr/bin/env python
"""Generate tests/golden/hook_cases.json by running the REFERENCE's model hooks (container only).

    python tests/golden/make_hook_golden.py

* LLaVA: upstream's own `LlavaMetaForCausalLM` (from /root/reference/llava) is subclassed with
  deterministic stand-ins for the vision side; the unpatched method gives the "plain" digests, the
  reference's `cus_prepare_inputs_labels_for_multimodal` the "compressed" ones.
* Qwen2.5-VL: the reference's `Qwen2_5_VLModel_forward` (written against transformers 4.5x) is run
  on a stand-in `self` that offers the few methods it calls; what it hands to the language model
  is recorded.
* Qwen2-VL: the reference hook cannot run against the current `vidcom2_compression` (it treats
  the returned rows as indices, see vidcom2_amd/models/qwen2_vl.py); no fixture, "parity unpinned".

Fixtures are data only (seeds, shapes, index lists, sha256 digests).
"""
from __future__ import annotations

import json
import os
import sys
import types
from types import SimpleNamespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

import _stub_models as S  # noqa: E402
from vidcom2_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
sha = synth.sha256_tensor

LLAVA_CASES = [
    dict(name="video_grid_bf16", merge="spatial_unpad", newline="grid", F=8, N=169, D=128, dt="bf16", seed=11,
         r="0.25", prefix=5, suffix=4),
    dict(name="video_grid_f32_r50", merge="spatial_unpad", newline="grid", F=6, N=169, D=64, dt="f32", seed=12,
         r="0.5", prefix=3, suffix=2),
    dict(name="ov_one_token_bf16", merge="spatial_unpad", newline="one_token", F=8, N=196, D=128, dt="bf16",
         seed=13, r="0.25", prefix=5, suffix=4),
    dict(name="ov_one_token_f16_r15", merge="spatial_unpad", newline="one_token", F=16, N=196, D=256, dt="f16",
         seed=14, r="0.15", prefix=2, suffix=6),
    # configurations the reference leaves untouched
    dict(name="frame_passthrough", merge="spatial_unpad", newline="frame", F=4, N=196, D=64, dt="f32", seed=15,
         r="0.25", prefix=2, suffix=2),
    dict(name="one_token_no_unpad_passthrough", merge="spatial", newline="one_token", F=4, N=196, D=64, dt="f32",
         seed=16, r="0.25", prefix=2, suffix=2),
]

QWEN25_CASES = [
    dict(name="one_video_bf16", D=128, dt="bf16", seed=21, r="0.25", grids=[[8, 12, 12]], prefix=15, between=0,
         suffix=9, mask="2d"),
    dict(name="two_videos_f32_r50_4dmask", D=64, dt="f32", seed=22, r="0.5", grids=[[4, 8, 8], [6, 12, 8]],
         prefix=7, between=3, suffix=5, mask="4d"),
    dict(name="one_video_f16_r10", D=256, dt="f16", seed=23, r="0.1", grids=[[16, 16, 16]], prefix=4, between=0,
         suffix=3, mask="none"),
]


QWEN3_CASES = [
    dict(name="video_only_bf16", D=128, dt="bf16", seed=31, r="0.25", grids=[[8, 12, 12]], prefix=11, between=0,
         suffix=6, mask="2d", n_image=0),
    dict(name="image_and_two_videos_f32", D=64, dt="f32", seed=32, r="0.4", grids=[[4, 8, 8], [6, 12, 8]], prefix=5,
         between=3, suffix=4, mask="4d", n_image=9),
]


def run_llava():
    from llava.model.llava_arch import LlavaMetaForCausalLM
    from token_compressor.vidcom2.models.llava import cus_prepare_inputs_labels_for_multimodal as ref_hook

    class RefLlava(LlavaMetaForCausalLM):
        def __init__(self, c):
            stub = S.StubLlava(c["D"], DT[c["dt"]], "cpu", c["seed"], c["merge"], c["newline"])
            self.config, self.model, self.device = stub.config, stub.model, stub.device

        def get_model(self):
            return self.model

        def encode_images(self, images):
            return images[..., 0]

        def get_2dPool(self, image_feature, stride=2):
            return image_feature

    out = []
    for c in LLAVA_CASES:
        os.environ["R_RATIO"] = c["r"]
        ids, feats = S.llava_inputs(c)
        m = RefLlava(c)
        plain = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)
        m.prepare_inputs_labels_for_multimodal = types.MethodType(ref_hook, m)
        comp = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)
        # our own stand-in must reproduce upstream's plain result (it is what the GPU tests wrap)
        stub = S.StubLlava(c["D"], DT[c["dt"]], "cpu", c["seed"], c["merge"], c["newline"])
        mine = stub.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)
        assert torch.equal(mine[4], plain[4]), c["name"]
        # and our wrapper hook bound on UPSTREAM's real class (device pass replaced by the oracle) must give what
        # the reference's patched copy gives
        import oracle as O
        import vidcom2_amd.models.llava as HL
        O.set_mode("torch")
        saved = HL.vidcom2_compression
        HL.vidcom2_compression = (lambda flat, model="llava_ov", base_scale=0.25, frame_token_len=None, img_feat=None:
                                  O.vidcom2_compression(flat.contiguous(), model, base_scale, frame_token_len, img_feat))
        try:
            m2 = RefLlava(c)
            m2.prepare_inputs_labels_for_multimodal = types.MethodType(HL.cus_prepare_inputs_labels_for_multimodal, m2)
            ours = m2.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)
        finally:
            HL.vidcom2_compression = saved
        assert torch.equal(ours[4], comp[4]), ("wrapper on upstream != reference hook", c["name"])
        out.append(dict(c, plain_shape=list(plain[4].shape), plain_sha=sha(plain[4]),
                        comp_shape=list(comp[4].shape), comp_sha=sha(comp[4])))
        print("llava", c["name"], list(plain[4].shape), "->", list(comp[4].shape))
    return out


def run_qwen25():
    import transformers.models.qwen2_5_vl.modeling_qwen2_5_vl as hf
    if not hasattr(hf, "is_torchdynamo_compiling"):        # name the reference imports from transformers 4.5x
        hf.is_torchdynamo_compiling = lambda: False
    from token_compressor.vidcom2.models.qwen2_5_vl import Qwen2_5_VLModel_forward as ref_forward

    out = []
    for c in QWEN25_CASES:
        os.environ["R_RATIO"], os.environ["COMPRESSOR"] = c["r"], "vidcom2"
        ids, feats, pos, mask = S.qwen_inputs(c)
        table = S.embed_table(c["D"], DT[c["dt"]], c["seed"])
        seen = {}

        class Self:
            config = SimpleNamespace(output_attentions=False, output_hidden_states=False, use_return_dict=True,
                                     video_token_id=S.VIDEO_ID, image_token_id=S.IMAGE_ID)
            visual = SimpleNamespace(spatial_merge_size=2)
            rope_deltas = None

            def get_input_embeddings(self):
                return lambda i: table[i]

            def get_video_features(self, pv, grid):
                return tuple(feats)

            def get_placeholder_mask(self, input_ids, inputs_embeds, image_features=None, video_features=None):
                im = (input_ids == S.IMAGE_ID).unsqueeze(-1).expand_as(inputs_embeds)
                vm = (input_ids == S.VIDEO_ID).unsqueeze(-1).expand_as(inputs_embeds)
                return im, vm

            def language_model(self, **kw):
                seen.update(kw)
                return SimpleNamespace(last_hidden_state=kw["inputs_embeds"], past_key_values=None)

        ref_forward(Self(), input_ids=ids, attention_mask=mask, position_ids=pos,
                    pixel_values_videos=torch.zeros(1, 4), video_grid_thw=torch.tensor(c["grids"]))
        keep = seen["position_ids"][0, 0].tolist()
        out.append(dict(c, seq_len=ids.shape[1], keep_token_indices=keep, embeds_sha=sha(seen["inputs_embeds"]),
                        mask_sha=None if mask is None else sha(seen["attention_mask"]),
                        mask_shape=None if mask is None else list(seen["attention_mask"].shape)))
        print("qwen2_5_vl", c["name"], ids.shape[1], "->", len(keep))
    return out


def run_qwen3():
    import transformers.models.qwen3_vl.modeling_qwen3_vl as hf
    if not hasattr(hf, "is_torchdynamo_compiling"):
        hf.is_torchdynamo_compiling = lambda: False
    from token_compressor.vidcom2.models.qwen3_vl import Qwen3VLModel_forward as ref_forward

    out = []
    for c in QWEN3_CASES:
        os.environ["R_RATIO"], os.environ["COMPRESSOR"] = c["r"], "vidcom2"
        dtype = S.DT[c["dt"]]
        ids, feats, pos, mask = S.qwen_inputs(c)
        n_vid = sum(f.shape[0] for f in feats)
        table = S.embed_table(c["D"], dtype, c["seed"])
        deep_v = S.deepstack_feats(n_vid, c["D"], dtype, c["seed"])
        img = S.video_feats(1, c["n_image"], c["D"], dtype, c["seed"] + 9)[0] if c["n_image"] else None
        deep_i = S.deepstack_feats(c["n_image"], c["D"], dtype, c["seed"] + 9) if c["n_image"] else None
        seen = {}

        class Self:
            config = SimpleNamespace(video_token_id=S.VIDEO_ID, image_token_id=S.IMAGE_ID)
            visual = SimpleNamespace(spatial_merge_size=2)
            rope_deltas = None

            def get_input_embeddings(self):
                return lambda i: table[i]

            def get_video_features(self, pv, grid):
                return tuple(feats), deep_v

            def get_image_features(self, pv, grid):
                return (img,), deep_i

            def get_placeholder_mask(self, input_ids, inputs_embeds, image_features=None, video_features=None):
                im = (input_ids == S.IMAGE_ID).unsqueeze(-1).expand_as(inputs_embeds)
                vm = (input_ids == S.VIDEO_ID).unsqueeze(-1).expand_as(inputs_embeds)
                return im, vm

            def language_model(self, **kw):
                seen.update(kw)
                return SimpleNamespace(last_hidden_state=kw["inputs_embeds"], past_key_values=None)

        ref_forward(Self(), input_ids=ids, attention_mask=mask, position_ids=pos,
                    pixel_values=torch.zeros(1, 4) if c["n_image"] else None,
                    image_grid_thw=torch.tensor([[1, 6, 6]]) if c["n_image"] else None,
                    pixel_values_videos=torch.zeros(1, 4), video_grid_thw=torch.tensor(c["grids"]))
        keep = seen["position_ids"][0, 0].tolist()
        out.append(dict(c, seq_len=ids.shape[1], keep_token_indices=keep, embeds_sha=sha(seen["inputs_embeds"]),
                        mask_sha=None if mask is None else sha(seen["attention_mask"]),
                        mask_shape=None if mask is None else list(seen["attention_mask"].shape),
                        vpm_sha=sha(seen["visual_pos_masks"].to(torch.uint8)),
                        deep_shapes=[list(d.shape) for d in seen["deepstack_visual_embeds"]],
                        deep_sha=[sha(d) for d in seen["deepstack_visual_embeds"]]))
        print("qwen3_vl", c["name"], ids.shape[1], "->", len(keep), out[-1]["deep_shapes"])
    return out


def main():
    cases = dict(llava=run_llava(), qwen2_5_vl=run_qwen25(), qwen3_vl=run_qwen3())
    with open(os.path.join(HERE, "hook_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote hook_cases.json")


if __name__ == "__main__":
    main()
