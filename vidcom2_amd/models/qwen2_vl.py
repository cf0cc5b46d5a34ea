"""Qwen2-VL hooks (reference: token_compressor/vidcom2/models/qwen2_vl.py:10-199).

The reference patches two methods of the transformers-4.4x layout:

* `Qwen2VL_ViT_forward(self, hidden_states, grid_thw)` on the vision tower: runs the tower and the
  merger, then `vidcom2_compression(merged, "qwen2_vl", base_scale=R_RATIO,
  frame_token_len=(h//2)*(w//2))`, returning `(merged, keep_index)` (qwen2_vl.py:10-44);
* `Qwen2VLGeneration_forward` on the generation model: scatters *all* video embeddings, computes
  rope positions on the full prompt, then keeps `[system tokens] + (keep_index + 15) + [the rest]`
  of `inputs_embeds`, `input_ids`, `position_ids`, `attention_mask` before the decoder runs
  (qwen2_vl.py:101-147); active whenever `pixel_values_videos` is passed, i.e. in prefill.

Status of the reference hook: it predates the current `vidcom2_compression`, which returns the
kept *rows* (vidcom2.py:36, 91), while the hook adds 15 to the result and uses it as positions;
it also hard-codes the 15-token system prompt of the Qwen2-VL chat template.  The evident intent
-- keep text plus the selected video positions -- is what is implemented here: `keep_index` is the
ascending index list of the kept video tokens, and positions are taken from the video placeholder
mask (identical to "15 + index" for that template, correct for any other prompt).

Three entry points:

* `Qwen2VL_ViT_forward`        -- same contract as the reference, wraps the tower's own forward;
* `Qwen2VLGeneration_forward`  -- transformers-4.4x layout (`self.visual`, `self.model` = decoder);
* `Qwen2VLModel_forward`       -- current transformers layout (`Qwen2VLModel.forward`, installed like
  the Qwen2.5-VL hook: `model.model.forward = types.MethodType(Qwen2VLModel_forward, model.model)`).
"""
from __future__ import annotations

import inspect

import torch
from torch import Tensor

from ..vidcom2 import _as_int, compress
from ._intercept import original_method, retention_ratio, shadow
from ._prefill_prune import PruneState, _prune_attention, run_with_pruning

__all__ = ["Qwen2VL_ViT_forward", "Qwen2VLGeneration_forward", "Qwen2VLModel_forward", "_keep_index"]


def _keep_index(merged: Tensor, grid_thw: Tensor, merge_size: int = 2) -> Tensor:
    """qwen2_vl.py:33-43: tokens per frame from the (single) video's grid, then one pass over all
    merged video tokens.  Like the reference's `int(frame_token_len)`, more than one video in the
    call is an error."""
    token_per_frame = (grid_thw[:, 1] // merge_size) * (grid_thw[:, 2] // merge_size)
    tpf = _as_int(token_per_frame)
    return compress(merged, tpf, retention_ratio(), "linear", gather=False).global_idx


def _merged_of(out):
    return out if torch.is_tensor(out) else out.pooler_output


def Qwen2VL_ViT_forward(self, hidden_states: Tensor, grid_thw: Tensor, **kwargs):
    original = original_method(self, "forward", Qwen2VL_ViT_forward)
    merged = _merged_of(original(hidden_states, grid_thw=grid_thw, **kwargs))
    merge_size = int(getattr(self, "spatial_merge_size", 2))
    return merged, _keep_index(merged, grid_thw, merge_size)


def Qwen2VLGeneration_forward(self, *args, **kwargs):
    """transformers-4.4x layout.  `self.visual` is expected to carry `Qwen2VL_ViT_forward` (it
    returns `(embeds, keep_index)`); the tower call is unwrapped for the model's own forward and
    the decoder call is intercepted to cut the prompt."""
    original = original_method(self, "forward", Qwen2VLGeneration_forward)
    try:
        bound = inspect.signature(original).bind_partial(*args, **kwargs).arguments
    except TypeError:
        bound = dict(kwargs)
    if bound.get("pixel_values_videos") is None or bound.get("input_ids") is None:
        with shadow(self, visual=_TowerShim(self.visual, None)):
            return original(*args, **kwargs)

    input_ids = bound["input_ids"]
    state = PruneState()
    tower = _TowerShim(self.visual, state)
    decoder = self.model

    class _DecoderShim:
        def __getattr__(self, name):
            return getattr(decoder, name)

        def __call__(self, *a, **kw):
            embeds = kw.get("inputs_embeds")
            if state.kept_video is not None and torch.is_tensor(embeds):
                vm = (input_ids == self_cfg.video_token_id)[0].to(embeds.device)
                video_pos = vm.nonzero(as_tuple=False).squeeze(-1)
                flags = ~vm
                flags[video_pos[state.kept_video]] = True
                keep = flags.nonzero(as_tuple=False).squeeze(-1)
                kw["inputs_embeds"] = embeds[:, keep, :]
                if torch.is_tensor(kw.get("position_ids")):
                    kw["position_ids"] = kw["position_ids"][..., keep]
                if "attention_mask" in kw:
                    kw["attention_mask"] = _prune_attention(kw["attention_mask"], keep)
                state.keep_token_indices, state.pruned = keep, True
            return decoder(*a, **kw)

    self_cfg = self.config
    with shadow(self, visual=tower, model=_DecoderShim()):
        out = original(*args, **kwargs)
    self.__dict__["_vidcom2_last"] = state
    return out


class _TowerShim:
    """`self.visual` for the duration of one legacy forward: hands the model plain embeddings and
    remembers the keep list of the video call (the call whose grid is `video_grid_thw`)."""

    def __init__(self, tower, state):
        self._tower, self._state = tower, state

    def __getattr__(self, name):
        return getattr(self._tower, name)

    def __call__(self, *args, **kwargs):
        out = self._tower(*args, **kwargs)
        if isinstance(out, tuple) and len(out) == 2 and torch.is_tensor(out[1]) and out[1].dtype == torch.int64:
            if self._state is not None:
                self._state.kept_video = out[1]
            return out[0]
        return out


def Qwen2VLModel_forward(
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
    **kwargs,
):
    """The parameter list of the reference's Qwen2-VL hook (models/qwen2_vl.py:46-64, there on the CausalLM and with
    `labels`; this one sits on the inner model like the other Qwen hooks)."""
    from .qwen2_5_vl import named_call_kwargs
    original = original_method(self, "forward", Qwen2VLModel_forward)
    named = dict(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                 past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache,
                 output_attentions=output_attentions, output_hidden_states=output_hidden_states, return_dict=return_dict,
                 pixel_values=pixel_values, pixel_values_videos=pixel_values_videos, image_grid_thw=image_grid_thw,
                 video_grid_thw=video_grid_thw, rope_deltas=rope_deltas, cache_position=cache_position)
    args, kwargs = (), named_call_kwargs(self, Qwen2VLModel_forward, named, kwargs)
    if pixel_values_videos is None or video_grid_thw is None:
        return original(*args, **kwargs)
    merge_size = int(getattr(self.visual, "spatial_merge_size", 2))

    def choose(video_embeds: Tensor):
        return _keep_index(video_embeds, video_grid_thw, merge_size)

    return run_with_pruning(self, lambda: original(*args, **kwargs), choose)
