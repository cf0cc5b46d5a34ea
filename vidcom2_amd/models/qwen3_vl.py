"""Qwen3-VL hook (reference: token_compressor/vidcom2/models/qwen3_vl.py:20-238).

    model.model.forward = types.MethodType(Qwen3VLModel_forward, model.model)      # README.md:86-93

Same gating, per-video scoring and prompt pruning as the Qwen2.5-VL hook (`qwen2_5_vl.py`); in addition
the deepstack features travel with the visual tokens, so `visual_pos_masks` and every layer of
`deepstack_visual_embeds` are cut to the kept positions (reference lines 141-149, 200-226) -- done by the
shared language-model interceptor (`_prefill_prune.py`), the installed transformers' own forward builds them.
"""
from __future__ import annotations

from .qwen2_5_vl import _compute_keep_indices, hooked_forward  # noqa: F401  (same scoring, re-exported)

__all__ = ["Qwen3VLModel_forward", "_compute_keep_indices"]


def Qwen3VLModel_forward(self, *args, **kwargs):
    return hooked_forward(self, Qwen3VLModel_forward, args, kwargs)
