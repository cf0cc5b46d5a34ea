"""Qwen2.5-VL hook (reference: token_compressor/vidcom2/models/qwen2_5_vl.py:20-202).

Install like the reference does for its Qwen hooks (README.md:86-93):

    model.model.forward = types.MethodType(Qwen2_5_VLModel_forward, model.model)

Behaviour kept from the reference:

* active only when `COMPRESSOR=vidcom2`, the call carries `pixel_values_videos` + `video_grid_thw`,
  the KV cache is empty (prefill) and the batch size is 1 (qwen2_5_vl.py:120-129); otherwise the
  model's own forward runs unchanged;
* every video is scored on its own (`_compute_keep_indices`, qwen2_5_vl.py:20-33, tokens per frame
  = h*w / merge_size**2), retention ratio from `R_RATIO`;
* the prompt is cut down to text + kept video positions: `inputs_embeds`, `attention_mask`
  (2-D or 4-D) and the *pre-computed* `position_ids` are sliced with the same index list
  (qwen2_5_vl.py:155-183).

The reference does this in a copy of an older `Qwen2_5_VLModel.forward`; here the installed
transformers' own forward runs and only its call into the language model is intercepted
(`_prefill_prune.py`), so the hook carries no upstream code and follows upstream changes.
"""
from __future__ import annotations

import inspect

import torch
from torch import Tensor

from ..vidcom2 import compress
from ._intercept import compressor_enabled, original_method, retention_ratio
from ._prefill_prune import run_with_pruning

__all__ = ["Qwen2_5_VLModel_forward", "_compute_keep_indices"]


def _compute_keep_indices(flat_features: Tensor, grid_thw: Tensor, spatial_merge_size: int,
                          base_scale: float) -> Tensor:
    """Kept token indices (ascending, int64) of one video -- qwen2_5_vl.py:20-33.  The reference
    chains the five stage functions; this is the same result from the fused device pass."""
    t, h, w = (int(v) for v in grid_thw.tolist())
    frame_tokens = (h * w) // (spatial_merge_size ** 2)
    if frame_tokens <= 0 or flat_features.numel() == 0:
        return torch.arange(flat_features.shape[0], device=flat_features.device)
    return compress(flat_features, frame_tokens, base_scale, "linear", gather=False).global_idx


def _cache_is_empty(past_key_values) -> bool:
    return past_key_values is None or past_key_values.get_seq_length() == 0


def _keep_per_video(video_embeds: Tensor, video_grid_thw: Tensor, merge_size: int, base_scale: float,
                    keep_fn) -> Tensor:
    split_sizes = (video_grid_thw.prod(-1) // merge_size ** 2).tolist()
    kept, offset = [], 0
    for grid, feat in zip(video_grid_thw, torch.split(video_embeds, split_sizes)):
        kept.append(keep_fn(feat, grid, merge_size, base_scale) + offset)
        offset += feat.shape[0]
    return torch.sort(torch.cat(kept)).values


def hooked_forward(self, hook, args, kwargs, keep_fn_name: str = "_compute_keep_indices"):
    """Shared by the Qwen2.5-VL and Qwen3-VL style hooks (same gating, per-video scoring)."""
    original = original_method(self, "forward", hook)
    try:
        bound = inspect.signature(original).bind_partial(*args, **kwargs).arguments
    except TypeError:
        bound = dict(kwargs)
    bound = {**bound.get("kwargs", {}), **bound}
    video_grid_thw = bound.get("video_grid_thw")
    active = (compressor_enabled() and bound.get("pixel_values_videos") is not None
              and video_grid_thw is not None and _cache_is_empty(bound.get("past_key_values")))
    if not active:
        return original(*args, **kwargs)

    merge_size = self.visual.spatial_merge_size
    base_scale = retention_ratio()
    keep_fn = globals()[keep_fn_name]

    def choose(video_embeds: Tensor):
        return _keep_per_video(video_embeds, video_grid_thw, merge_size, base_scale, keep_fn)

    return run_with_pruning(self, lambda: original(*args, **kwargs), choose)


def named_call_kwargs(self, hook, named: dict, extra: dict) -> dict:
    """The hooks carry the reference's EXPLICIT parameter lists (inspect.signature consumers -- HF generate's argument
    validation, tracing tools -- see the names they expect); the installed transformers' forward is then called by
    NAME.  A reference-era parameter the installed forward no longer has (e.g. `return_dict`, `second_per_grid_ts`) is
    dropped when it was left at None and passed through otherwise (the model's own TypeError is the right answer)."""
    original = original_method(self, "forward", hook)
    params = inspect.signature(original).parameters
    open_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())
    out = {k: v for k, v in named.items() if v is not None or k in params}
    if not open_kw:
        out = {k: v for k, v in out.items() if k in params or v is not None}
    out.update(extra)
    return out


def Qwen2_5_VLModel_forward(
    self,
    input_ids=None,
    attention_mask=None,
    position_ids=None,
    past_key_values=None,
    inputs_embeds=None,
    use_cache=None,
    output_attentions=None,
    output_hidden_states=None,
    return_dict=None,
    pixel_values=None,
    pixel_values_videos=None,
    image_grid_thw=None,
    video_grid_thw=None,
    rope_deltas=None,
    cache_position=None,
    second_per_grid_ts=None,
    **kwargs,
):
    """Same parameter list as the reference's hook (models/qwen2_5_vl.py:36-55)."""
    named = dict(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                 past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache,
                 output_attentions=output_attentions, output_hidden_states=output_hidden_states, return_dict=return_dict,
                 pixel_values=pixel_values, pixel_values_videos=pixel_values_videos, image_grid_thw=image_grid_thw,
                 video_grid_thw=video_grid_thw, rope_deltas=rope_deltas, cache_position=cache_position,
                 second_per_grid_ts=second_per_grid_ts)
    return hooked_forward(self, Qwen2_5_VLModel_forward, (),
                          named_call_kwargs(self, Qwen2_5_VLModel_forward, named, kwargs))
