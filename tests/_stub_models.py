"""Tiny deterministic host models for the hook tests (test infrastructure).

The hooks wrap a model's *own* methods, so the tests need models that have them:

* `StubLlava` -- a compact batch-1 model of the LLaVA-NeXT multimodal input preparation
  (vision features -> per-video newline handling -> spliced into the text embeddings).  It is
  checked against upstream itself: `tests/golden/make_hook_golden.py` runs the real
  `LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal` on the same synthetic inputs and
  `hook_cases.json` holds its digests ("plain"), next to the reference hook's ("compressed").
* `make_qwen_vl_model` -- the installed transformers' real `Qwen2_5_VLModel` / `Qwen2VLModel`
  with a tiny config, its video tower output replaced by synthetic features and its language
  model replaced by a recorder, so the test sees exactly what the decoder would be fed.

All tensors come from `vidcom2_amd.synth` (bit-portable), so digests match across machines.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from vidcom2_amd import synth

IMAGE_TOKEN_INDEX = -200        # llava.constants
VOCAB = 64
_TID_EMBED, _TID_NEWLINE = 40, 41


def embed_table(D: int, dtype, seed: int) -> torch.Tensor:
    return synth.to_torch(synth.gauss(seed, _TID_EMBED, 0, VOCAB * D).reshape(VOCAB, D), dtype)


def newline_vec(D: int, dtype, seed: int) -> torch.Tensor:
    return synth.to_torch(synth.gauss(seed, _TID_NEWLINE, 0, D), dtype)


def video_feats(F: int, N: int, D: int, dtype, seed: int) -> torch.Tensor:
    """[F, N, D] pooled per-frame features of one clip."""
    return synth.make(F, N, D, dtype, seed, "drift").reshape(F, N, D)


def text_ids(n: int, seed: int) -> torch.Tensor:
    g = synth.gauss(seed, 42, 0, n)
    return torch.from_numpy((np.abs(g) * 1000).astype(np.int64) % 50 + 1)      # ids 1..50


# ------------------------------------------------------------------------------------------
# LLaVA
# ------------------------------------------------------------------------------------------
class _Inner:
    def __init__(self, D, dtype, device, seed):
        self._table = embed_table(D, dtype, seed).to(device)
        self.image_newline = newline_vec(D, dtype, seed).to(device)
        self._tower = SimpleNamespace(num_patches_per_side=27)

    def embed_tokens(self, ids):
        return self._table[ids]

    def get_vision_tower(self):
        return self._tower


class StubLlava:
    """`images` carry already-projected, already-pooled features: a list of `[F, P, D, 1]` tensors
    (4-D like a stack of frames; `encode_images` drops the trailing axis)."""

    def __init__(self, D, dtype, device, seed, merge_type="spatial_unpad", newline="one_token"):
        self.config = SimpleNamespace(mm_patch_merge_type=merge_type, mm_newline_position=newline,
                                      image_aspect_ratio="square")
        self.model = _Inner(D, dtype, device, seed)
        self.device = torch.device(device)

    # -- the pieces upstream's method is built from -------------------------------------------
    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.model.get_vision_tower()

    def encode_images(self, images):
        return images[..., 0]

    def get_2dPool(self, image_feature, stride=2):
        return image_feature

    def add_token_per_grid(self, image_feature):
        F, P, D = image_feature.shape
        h = int(round(P ** 0.5))
        rows = image_feature.reshape(F, h, h, D)
        nl = self.model.image_newline.to(rows.device).expand(F, h, 1, D)
        return torch.cat((rows, nl), dim=2).reshape(F * h * (h + 1), D)

    def add_token_per_frame(self, image_feature):
        F, P, D = image_feature.shape
        nl = self.model.image_newline.to(image_feature.device).expand(F, 1, D)
        return torch.cat((image_feature, nl), dim=1)

    # -- batch-1 model of prepare_inputs_labels_for_multimodal ---------------------------------
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values,
                                             labels, images, modalities=["image"], image_sizes=None):
        if images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        if isinstance(modalities, str):
            modalities = [modalities]
        assert type(images) is list and input_ids.shape[0] == 1 and all(im.ndim == 4 for im in images)
        videos = {i for i, m in enumerate(modalities) if m == "video"}
        sizes = [im.shape[0] for im in images]
        encoded = torch.split(self.encode_images(torch.cat(images, dim=0)), sizes)
        feats = [self.get_2dPool(f) if i in videos else f for i, f in enumerate(encoded)]

        merge, newline = self.config.mm_patch_merge_type, self.config.mm_newline_position
        if merge == "flat":
            feats = [f.flatten(0, 1) for f in feats]
        else:
            assert merge.startswith("spatial")
            done = []
            for i, f in enumerate(feats):
                if i in videos:
                    if newline == "grid":
                        done.append(self.add_token_per_grid(f))
                    elif newline == "frame":
                        done.append(self.add_token_per_frame(f).flatten(0, 1))
                    elif newline == "one_token":
                        f = f.flatten(0, 1)
                        if "unpad" in merge:
                            f = torch.cat((f, self.model.image_newline[None].to(f.device)), dim=0)
                        done.append(f)
                    elif newline == "no_token":
                        done.append(f.flatten(0, 1))
                    else:
                        raise ValueError(f"Unexpected mm_newline_position: {newline}")
                else:
                    assert f.shape[0] == 1, "stub handles single-patch images only"
                    f = f[0]
                    if "unpad" in merge:
                        f = torch.cat((f, self.model.image_newline[None]), dim=0)
                    done.append(f)
            feats = done

        ids = input_ids[0]
        pieces, k, start = [], 0, 0
        marks = (ids == IMAGE_TOKEN_INDEX).nonzero(as_tuple=False).squeeze(-1).tolist()
        for m in marks + [ids.shape[0]]:
            if m > start:
                pieces.append(self.model.embed_tokens(ids[start:m]))
            if m < ids.shape[0]:
                pieces.append(feats[k])
                k += 1
            start = m + 1
        embeds = torch.cat(pieces, dim=0)[None]
        return None, position_ids, attention_mask, past_key_values, embeds, labels


# ------------------------------------------------------------------------------------------
# Qwen2-VL / Qwen2.5-VL (real transformers classes, tiny config)
# ------------------------------------------------------------------------------------------
IMAGE_ID, VIDEO_ID = 60, 61


class Recorder(torch.nn.Module):
    """Stands where the decoder stack is; remembers its keyword arguments."""

    def __init__(self, table: torch.Tensor):
        super().__init__()
        self.embed = torch.nn.Embedding.from_pretrained(table, freeze=True)
        self.calls = []

    def get_input_embeddings(self):
        return self.embed

    def forward(self, **kwargs):
        from transformers.modeling_outputs import BaseModelOutputWithPast
        self.calls.append(kwargs)
        return BaseModelOutputWithPast(last_hidden_state=kwargs["inputs_embeds"], past_key_values=kwargs.get(
            "past_key_values"))


def make_qwen_vl_model(family: str, D: int, dtype, device, seed: int):
    """`family` in {"qwen2_5_vl", "qwen2_vl", "qwen3_vl"} -> (model, recorder)."""
    if family == "qwen2_5_vl":
        from transformers.models.qwen2_5_vl import Qwen2_5_VLConfig as Cfg
        from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLModel as Model
        vision = dict(depth=1, hidden_size=32, intermediate_size=32, num_heads=2, out_hidden_size=D,
                      fullatt_block_indexes=[0], spatial_merge_size=2)
    elif family == "qwen3_vl":
        from transformers.models.qwen3_vl import Qwen3VLConfig as Cfg
        from transformers.models.qwen3_vl.modeling_qwen3_vl import Qwen3VLModel as Model
        vision = dict(depth=1, hidden_size=32, intermediate_size=32, num_heads=2, out_hidden_size=D,
                      spatial_merge_size=2, deepstack_visual_indexes=[0], num_position_embeddings=16)
    else:
        from transformers.models.qwen2_vl import Qwen2VLConfig as Cfg
        from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLModel as Model
        vision = dict(depth=1, embed_dim=32, hidden_size=D, num_heads=2, mlp_ratio=1, spatial_merge_size=2)
    text = dict(vocab_size=VOCAB, hidden_size=D, intermediate_size=32, num_hidden_layers=1,
                num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=4096, bos_token_id=1,
                eos_token_id=2, pad_token_id=0,
                rope_parameters=dict(rope_type="default", mrope_section=[D // 16, D // 16 + D // 32,
                                                                         D // 4 - D // 16 - (D // 16 + D // 32)],
                                     rope_theta=10000.0))
    cfg = Cfg(text_config=text, vision_config=vision, image_token_id=IMAGE_ID, video_token_id=VIDEO_ID)
    with torch.device("meta"):
        model = Model(cfg)
    model.language_model = None
    rec = Recorder(embed_table(D, dtype, seed).to(device))
    model.language_model = rec
    model.eval()
    return model, rec


def set_video_features(model, feats_per_video, deepstack=None):
    """Make the model's video tower deliver `feats_per_video` (list of [n_i, D]) and, for Qwen3-VL, the
    per-layer deepstack features (list of [sum n_i, D])."""
    out = SimpleNamespace(pooler_output=tuple(feats_per_video), deepstack_features=deepstack)
    model.__dict__["get_video_features"] = lambda *a, **k: out


def set_image_features(model, feats_per_image, deepstack=None):
    out = SimpleNamespace(pooler_output=tuple(feats_per_image), deepstack_features=deepstack)
    model.__dict__["get_image_features"] = lambda *a, **k: out


def deepstack_feats(n: int, D: int, dtype, seed: int, layers: int = 2):
    """Synthetic per-layer deepstack features [n, D] (stream ids 44+)."""
    return [synth.to_torch(synth.gauss(seed, 44 + l, 0, n * D).reshape(n, D), dtype) for l in range(layers)]


def qwen_prompt(n_prefix: int, video_lens, n_between: int, n_suffix: int, seed: int, n_image: int = 0) -> torch.Tensor:
    parts = [text_ids(n_prefix, seed)]
    if n_image:
        parts += [torch.full((n_image,), IMAGE_ID, dtype=torch.int64), text_ids(2, seed + 300)]
    for i, n in enumerate(video_lens):
        if i:
            parts.append(text_ids(n_between, seed + 100 + i))
        parts.append(torch.full((n,), VIDEO_ID, dtype=torch.int64))
    parts.append(text_ids(n_suffix, seed + 7))
    return torch.cat(parts)[None]


# ------------------------------------------------------------------------------------------
# inputs of the fixture cases (tests/golden/hook_cases.json)
# ------------------------------------------------------------------------------------------
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def llava_inputs(c):
    ids = torch.cat((text_ids(c["prefix"], c["seed"]), torch.tensor([IMAGE_TOKEN_INDEX]),
                     text_ids(c["suffix"], c["seed"] + 1)))[None]
    feats = video_feats(c["F"], c["N"], c["D"], DT[c["dt"]], c["seed"])
    return ids, feats[..., None]


def qwen_inputs(c):
    dtype = DT[c["dt"]]
    feats, lens = [], []
    for i, (t, h, w) in enumerate(c["grids"]):
        n = (h * w) // 4
        feats.append(video_feats(t, n, c["D"], dtype, c["seed"] + i).reshape(t * n, c["D"]))
        lens.append(t * n)
    ids = qwen_prompt(c["prefix"], lens, c["between"], c["suffix"], c["seed"], c.get("n_image", 0))
    L = ids.shape[1]
    pos = torch.arange(L).view(1, 1, L).expand(3, 1, L).contiguous()
    if c["mask"] == "2d":
        mask = torch.ones(1, L, dtype=torch.int64)
    elif c["mask"] == "4d":
        mask = torch.triu(torch.full((L, L), float("-inf")), diagonal=1)[None, None].to(dtype)
    else:
        mask = None
    return ids, feats, pos, mask


def msg_inputs(c):
    """x [F, N, C] and center ([1, 1, C] | [F, 1, C]) of a `msg_cases.json` case; scaled so that the
    squared distances are O(1) and the Gaussians do not all underflow."""
    F, N, C = c["F"], c["N"], c["C"]
    s = np.float32(0.7 / np.sqrt(C))
    x = synth.to_torch(synth.make_fp32(F, N, C, c["seed"], "iid") * s, DT[c["dt"]])
    nc = 1 if c["kind"] == "video" else F
    cen = synth.to_torch(synth.make_fp32(nc, 1, C, c["seed"] + 50, "iid") * s, DT[c["dt"]])
    return x, cen
