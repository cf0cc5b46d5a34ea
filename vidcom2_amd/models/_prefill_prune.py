"""Prefill pruning shared by the Qwen-VL hooks.

What the reference's patched forwards do between "video embeddings scattered into the prompt"
and "language model called" (models/qwen2_5_vl.py:126-185, models/qwen2_vl.py:101-147), as an
interceptor around the model's own forward:

1. `get_placeholder_mask` is watched to learn where the video placeholders sit and which
   embeddings were scattered there (the model's own call, nothing recomputed);
2. the call into `self.language_model` is intercepted: kept video positions are decided by
   `choose(video_embeds)`, then `inputs_embeds`, `attention_mask` (2-D or 4-D) and
   `position_ids` are sliced to text + kept video positions and the real language model runs.

Position ids are computed by the model before the language-model call, i.e. on the *unpruned*
sequence, which is what the reference does on purpose (qwen2_5_vl.py:89: "Pre-compute position
ids before pruning so we can safely slice them later").
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from ._intercept import shadow


class PruneState:
    """Filled while the wrapped forward runs; read by tests through `model._vidcom2_last`."""

    def __init__(self) -> None:
        self.video_mask: Optional[torch.Tensor] = None       # [B, S] bool
        self.video_embeds: Optional[torch.Tensor] = None     # [n_video_tokens, D]
        self.kept_video: Optional[torch.Tensor] = None       # sorted indices into the video tokens
        self.keep_token_indices: Optional[torch.Tensor] = None   # sorted positions in the sequence
        self.pruned = False


def _prune_attention(attn, keep: torch.Tensor):
    if not torch.is_tensor(attn):
        return attn
    if attn.dim() == 2:
        return attn[:, keep]
    if attn.dim() == 4:
        return attn[:, :, keep, :][:, :, :, keep]
    return attn


def _fusable(embeds, kept, video_embeds, deep) -> bool:
    """The device kernels take one GPU, fp32 / bf16 / fp16 rows and (deep-stack) up to 8 same-shaped tensors."""
    if not (embeds.is_cuda and torch.is_tensor(kept) and kept.is_cuda and torch.is_tensor(video_embeds)):
        return False
    if embeds.dtype not in (torch.float32, torch.bfloat16, torch.float16) or embeds.dim() != 3:
        return False
    if deep is not None:
        deep = list(deep)
        if len(deep) > 8 or any((not torch.is_tensor(d)) or d.dim() != 2 or d.device != embeds.device
                                or d.dtype != deep[0].dtype or d.shape != deep[0].shape for d in deep):
            return False
        if deep and deep[0].dtype not in (torch.float32, torch.bfloat16, torch.float16):
            return False
    return True


class _LanguageModelShim:
    """Stands in for `self.language_model` during one forward; everything but the call itself
    (attributes, sub-modules, config) is the real module's."""

    def __init__(self, real, state: PruneState, choose: Callable[[torch.Tensor], Optional[torch.Tensor]]):
        object.__setattr__(self, "_real", real)
        object.__setattr__(self, "_state", state)
        object.__setattr__(self, "_choose", choose)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_real"), name)

    def __call__(self, *args, **kwargs):
        st: PruneState = self._state
        embeds = kwargs.get("inputs_embeds")
        if st.video_mask is not None and torch.is_tensor(embeds) and embeds.shape[0] == 1:
            kept = self._choose(st.video_embeds)
            if kept is not None:
                vm = st.video_mask[0].to(embeds.device)
                vpm = kwargs.get("visual_pos_masks")
                deep = kwargs.get("deepstack_visual_embeds") if torch.is_tensor(vpm) else None
                fused = _fusable(embeds, kept, st.video_embeds, deep) and vm.numel() == embeds.shape[1]
                vis_rows = None
                if fused:
                    # device-side keep list + ONE launch for text rows and kept video rows (vc2_keep_positions,
                    # vc2_gather_scatter): no nonzero() round trips
                    from ..fused import gather_scatter, keep_positions
                    n_video = int(st.video_embeds.shape[0])
                    want_vis = deep is not None and len(deep) > 0
                    keep, vis_rows = keep_positions(vm, kept, n_video, vpm[0].to(embeds.device) if want_vis else None,
                                                    int(deep[0].shape[0]) if want_vis else None)
                    keep_flags = None
                    # (no second read-back: keep_positions has just validated the list -- strictly ascending positions of
                    #  a mask as long as `embeds` -- and raised otherwise; vis_rows likewise against the deep-stack rows)
                    kwargs["inputs_embeds"] = gather_scatter([embeds[0]], keep, check=False)[0][None]
                else:
                    video_pos = vm.nonzero(as_tuple=False).squeeze(-1)
                    keep_flags = ~vm
                    keep_flags[video_pos[kept]] = True
                    keep = keep_flags.nonzero(as_tuple=False).squeeze(-1)
                    kwargs["inputs_embeds"] = embeds[:, keep, :]
                if torch.is_tensor(kwargs.get("input_ids")):
                    kwargs["input_ids"] = kwargs["input_ids"][:, keep]
                if "attention_mask" in kwargs:
                    am = kwargs["attention_mask"]
                    kwargs["attention_mask"] = ({k: _prune_attention(v, keep) for k, v in am.items()}
                                                if isinstance(am, dict) else _prune_attention(am, keep))
                pos = kwargs.get("position_ids")
                if torch.is_tensor(pos):
                    pos = pos[..., keep.to(pos.device)]
                    if pos.dim() == 3 and pos.shape[0] == 4:
                        # transformers >= 4.5x prepends a row of TEXT positions that the decoder only uses to
                        # detect packed sequences (any step != 1 starts a new one).  The gaps left by the dropped
                        # video tokens must not read as packing: the kept prompt is one sequence.
                        pos = pos.clone()
                        pos[0] = torch.arange(pos.shape[-1], device=pos.device, dtype=pos.dtype).expand_as(pos[0])
                    kwargs["position_ids"] = pos
                elif pos is None and "position_ids" in kwargs:
                    # the model could not build 3-D positions and leaves them to the decoder, which would number
                    # the PRUNED prompt 0..S'-1.  The reference computes positions before pruning and slices them
                    # (qwen2_5_vl.py:89-118): do the same with the plain positions of the full prompt.
                    rope = keep.view(1, 1, -1).expand(3, embeds.shape[0], -1)
                    text = torch.arange(keep.numel(), device=keep.device).view(1, 1, -1).expand(1, embeds.shape[0], -1)
                    kwargs["position_ids"] = torch.cat((text, rope), dim=0)
                cp = kwargs.get("cache_position")
                if torch.is_tensor(cp) and cp.dim() == 1 and cp.numel() == embeds.shape[1]:
                    kwargs["cache_position"] = cp[: keep.numel()]       # prefill: slots 0 .. S'-1 of the KV cache
                # Qwen3-VL deepstack: per-layer features of the visual tokens, row i <-> i-th True of the mask
                # (models/qwen3_vl.py:141-149, 200-226 of the reference)
                if torch.is_tensor(vpm):
                    if vis_rows is not None:
                        # the deep-stack tensors share ONE index list: one launch for all of them
                        kwargs["deepstack_visual_embeds"] = gather_scatter(list(deep), vis_rows, check=False)
                    else:
                        if keep_flags is None:
                            keep_flags = torch.zeros(vm.numel(), dtype=torch.bool, device=keep.device)
                            keep_flags[keep] = True
                        rows = keep_flags[vpm[0].to(keep_flags.device)]
                        deep_any = kwargs.get("deepstack_visual_embeds")
                        if deep_any is not None:
                            kwargs["deepstack_visual_embeds"] = [d[rows.to(d.device)] for d in deep_any]
                    kwargs["visual_pos_masks"] = vpm[:, keep.to(vpm.device)]
                st.kept_video, st.keep_token_indices, st.pruned = kept, keep, True
        return self._real(*args, **kwargs)


def run_with_pruning(model, forward: Callable, choose: Callable[[torch.Tensor], Optional[torch.Tensor]],
                     lm_attr: str = "language_model"):
    """Run `forward()` with the interceptors in place.  `choose(video_embeds)` returns the sorted
    kept indices into the video token list (or None to leave the prompt alone)."""
    state = PruneState()
    mask_inner = model.get_placeholder_mask

    def get_placeholder_mask(*args, **kwargs):
        out = mask_inner(*args, **kwargs)
        feats = kwargs.get("video_features")
        if feats is not None:
            vmask = out[1]
            st_mask = vmask[..., 0] if vmask.dim() == 3 else vmask
            state.video_mask, state.video_embeds = st_mask.bool(), feats
        return out

    shim = _LanguageModelShim(getattr(model, lm_attr), state, choose)
    with shadow(model, get_placeholder_mask=get_placeholder_mask, **{lm_attr: shim}):
        out = forward()
    model.__dict__["_vidcom2_last"] = state
    return out
