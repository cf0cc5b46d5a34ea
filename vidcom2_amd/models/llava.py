"""LLaVA-OneVision / LLaVA-Video hook (reference: token_compressor/vidcom2/models/llava.py:52-379).

Install exactly like the reference (README.md:76-83, lmms_eval/models/llava_onevision.py:157-163):

    model.prepare_inputs_labels_for_multimodal = types.MethodType(
        cus_prepare_inputs_labels_for_multimodal, model)

The reference's function is a copy of `LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal`
with two inserted calls, both in the per-video branch of the "spatial*" merge types:

* `mm_newline_position == "grid"` (LLaVA-Video; llava.py:114-127): the video's pooled features
  `[F, 169, D]` are flattened, `add_token_per_grid` builds the newline-augmented `[F*13*14, D]`
  stream, and `vidcom2_compression(flattened, "llava_vid", img_feat=stream)` replaces the stream
  by its kept rows;
* `mm_newline_position == "one_token"` with an "unpad" merge type (LLaVA-OneVision;
  llava.py:153-161): `image_feature.flatten(0, 1)` is replaced by
  `vidcom2_compression(image_feature.flatten(0, 1))` (model "llava_ov", 196 tokens per frame).

This wrapper runs the installed class's own method and intercepts exactly those two points:
`self.add_token_per_grid` (grid) and, for one_token, the tensor `self.get_2dPool` hands to the
branch, whose `flatten(0, 1)` yields the compressed rows.  Other configurations ("frame",
"no_token", "flat" merge, non-video inputs) pass through untouched, as in the reference.
"""
from __future__ import annotations

import torch

from ..vidcom2 import vidcom2_compression
from ._intercept import original_method, retention_ratio, shadow

__all__ = ["cus_prepare_inputs_labels_for_multimodal"]


class _NewlineFused(torch.Tensor):
    """The K kept rows as a view of a `[K + 1, D]` buffer whose last row already holds the model's newline
    embedding (written by the gather launch, `vc2_compress_tail`).  The caller's next statement is
    `torch.cat((rows, self.model.image_newline[None].to(rows.device)), dim=0)` (models/llava.py:160-168 of the
    reference): that exact call returns the buffer -- no second copy of the kept rows.  Anything else sees a plain
    tensor."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.cat and args and isinstance(args[0], (tuple, list)) and len(args[0]) == 2:
            rows, nl = args[0]
            dim = kwargs.get("dim", args[1] if len(args) > 1 else 0)
            fused = getattr(rows, "_vc2_fused", None)
            if (isinstance(rows, cls) and fused is not None and dim == 0 and torch.is_tensor(nl)
                    and not isinstance(nl, cls) and nl.shape == (1, rows.shape[1]) and nl.dtype == rows.dtype
                    and nl.device == rows.device and nl.data_ptr() == fused[1]):
                return fused[0]
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return out.as_subclass(torch.Tensor) if isinstance(out, cls) else out


class _CompressOnFlatten(torch.Tensor):
    """Pooled video features `[F, N, D]` whose `.flatten(0, 1)` is the compressed token list."""

    __torch_function__ = torch._C._disabled_torch_function_impl   # every op returns plain tensors

    def flatten(self, *args, **kwargs):
        plain = self.as_subclass(torch.Tensor)
        flat = plain.flatten(*args, **kwargs)
        if args == (0, 1) and not kwargs and plain.dim() == 3:
            newline = getattr(self, "_vc2_newline", None)
            if newline is not None and flat.is_cuda and newline.device == flat.device and newline.dtype == flat.dtype \
                    and newline.dim() == 1 and newline.shape[0] == flat.shape[1] and newline.is_contiguous():
                # kept rows + the newline row behind them in ONE gather launch
                from ..vidcom2 import MODEL_SPECS, compress
                res = compress(flat, MODEL_SPECS["llava_ov"]["tpf"], retention_ratio(), tail=newline[None],
                               stats_ws=getattr(self, "_vc2_stats_ws", None))
                rows = res.rows[: res.K].as_subclass(_NewlineFused)
                rows._vc2_fused = (res.rows, newline.data_ptr())
                return rows
            ws = getattr(self, "_vc2_stats_ws", None)
            if ws is not None and flat.is_cuda:
                from ..vidcom2 import MODEL_SPECS, compress
                return compress(flat, MODEL_SPECS["llava_ov"]["tpf"], retention_ratio(), stats_ws=ws).rows
            return vidcom2_compression(flat, base_scale=retention_ratio())
        return flat


class _PooledWithStats(torch.Tensor):
    """A pooled video `[F, N, D]` that carries the workspace holding its sweep-1 statistics (fused.pool_stats)."""

    __torch_function__ = torch._C._disabled_torch_function_impl


def _fused_pool(model, pool_inner, image_feature, args, kwargs):
    """get_2dPool + sweep 1 in one kernel when the configuration allows a bit-identical pooled tensor (average / max
    / bilinear with a feature size that is a multiple of torch's vector width).  None: not applicable.
    VC2_FUSED_POOL=off keeps the model's own pooling (the fused one reproduces torch's x86 CPU bits, which differ
    by rounding from what the model's pooling computes on the GPU)."""
    import os
    stride = args[0] if args else kwargs.get("stride", 2)
    mode = getattr(model.config, "mm_spatial_pool_mode", None)
    knob = os.getenv("VC2_FUSED_POOL", "")
    if knob == "off":
        return None
    if stride != 2 or mode not in ("average", "max", "bilinear") or not (torch.is_tensor(image_feature)
                                                                         and image_feature.is_cuda):
        return None
    if mode == "bilinear" and image_feature.shape[-1] % (8 if image_feature.dtype == torch.float32 else 16) != 0:
        return None
    if image_feature.dim() != 3 or image_feature.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return None
    try:
        side = int(model.get_vision_tower().num_patches_per_side)
    except Exception:
        return None
    if side * side != image_feature.shape[1]:
        return None
    if mode == "bilinear" and 2 * ((side + 1) // 2) > 128:
        return None          # torch's other bilinear kernel (out_h + out_w > 128): another summation order
    from ..fused import pool_stats
    out, ws = pool_stats(image_feature, side, side, mode)
    return out, ws


def cus_prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values,
                                             labels, images, modalities=["image"], image_sizes=None):
    original = original_method(self, "prepare_inputs_labels_for_multimodal",
                               cus_prepare_inputs_labels_for_multimodal)
    cfg = self.config
    merge_type = getattr(cfg, "mm_patch_merge_type", "flat")
    newline = getattr(cfg, "mm_newline_position", "one_token")
    per_sample = type(images) is list or (torch.is_tensor(images) and images.ndim == 5)

    patches = {}
    if per_sample and merge_type.startswith("spatial"):
        if newline == "grid":
            grid_inner = self.add_token_per_grid
            faster = bool(getattr(cfg, "add_faster_video", False))
            calls = [0]

            def add_token_per_grid(image_feature):
                out = grid_inner(image_feature)
                calls[0] += 1
                if faster and calls[0] % 2 == 0:
                    # second call per video builds the "faster" stream, which the reference leaves
                    # uncompressed (llava.py:130-131)
                    return out
                flat = image_feature.reshape(-1, image_feature.shape[-1])
                ws = getattr(image_feature, "_vc2_stats_ws", None)
                if ws is not None and flat.is_cuda:
                    from ..vidcom2 import MODEL_SPECS, compress
                    spec = MODEL_SPECS["llava_vid"]
                    return compress(flat, spec["tpf"], retention_ratio(), "grid_vid", spec["grid"], out,
                                    stats_ws=ws).rows
                return vidcom2_compression(flat, "llava_vid", base_scale=retention_ratio(), img_feat=out)

            patches["add_token_per_grid"] = add_token_per_grid
            pool_inner_g = self.get_2dPool

            def get_2dPool_grid(image_feature, *args, **kwargs):
                fused = _fused_pool(self, pool_inner_g, image_feature, args, kwargs)
                if fused is None:
                    return pool_inner_g(image_feature, *args, **kwargs)
                out = fused[0].as_subclass(_PooledWithStats)
                out._vc2_stats_ws = fused[1]
                return out

            patches["get_2dPool"] = get_2dPool_grid
        elif newline == "one_token" and "unpad" in merge_type:
            pool_inner = self.get_2dPool

            inner_model = getattr(self, "model", None)
            newline_vec = getattr(inner_model, "image_newline", None) if inner_model is not None else None

            def get_2dPool(image_feature, *args, **kwargs):
                fused = _fused_pool(self, pool_inner, image_feature, args, kwargs)
                if fused is None:
                    out = pool_inner(image_feature, *args, **kwargs).as_subclass(_CompressOnFlatten)
                else:
                    out = fused[0].as_subclass(_CompressOnFlatten)
                    out._vc2_stats_ws = fused[1]
                if torch.is_tensor(newline_vec):
                    out._vc2_newline = newline_vec.detach()
                return out

            patches["get_2dPool"] = get_2dPool

    with shadow(self, **patches):
        return original(input_ids, position_ids, attention_mask, past_key_values, labels, images,
                        modalities, image_sizes)
