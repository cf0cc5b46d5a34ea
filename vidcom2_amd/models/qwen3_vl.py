"""Qwen3-VL hook (reference: token_compressor/vidcom2/models/qwen3_vl.py:20-238).

    model.model.forward = types.MethodType(Qwen3VLModel_forward, model.model)      # README.md:86-93

Same gating, per-video scoring and prompt pruning as the Qwen2.5-VL hook (`qwen2_5_vl.py`); in addition
the deepstack features travel with the visual tokens, so `visual_pos_masks` and every layer of
`deepstack_visual_embeds` are cut to the kept positions (reference lines 141-149, 200-226) -- done by the
shared language-model interceptor (`_prefill_prune.py`), the installed transformers' own forward builds them.
"""
from __future__ import annotations

from .qwen2_5_vl import _compute_keep_indices, hooked_forward, named_call_kwargs  # noqa: F401  (same scoring)

__all__ = ["Qwen3VLModel_forward", "_compute_keep_indices"]


def Qwen3VLModel_forward(
    self,
    input_ids=None,
    attention_mask=None,
    position_ids=None,
    past_key_values=None,
    inputs_embeds=None,
    pixel_values=None,
    pixel_values_videos=None,
    image_grid_thw=None,
    video_grid_thw=None,
    cache_position=None,
    **kwargs,
):
    """Same parameter list as the reference's hook (models/qwen3_vl.py:36-48)."""
    named = dict(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                 past_key_values=past_key_values, inputs_embeds=inputs_embeds, pixel_values=pixel_values,
                 pixel_values_videos=pixel_values_videos, image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw,
                 cache_position=cache_position)
    return hooked_forward(self, Qwen3VLModel_forward, (), named_call_kwargs(self, Qwen3VLModel_forward, named, kwargs))
