"""Model hooks: the reference's `token_compressor/vidcom2/models/` surface.

    from vidcom2_amd.models.llava import cus_prepare_inputs_labels_for_multimodal
    from vidcom2_amd.models.qwen2_5_vl import Qwen2_5_VLModel_forward
    from vidcom2_amd.models.qwen2_vl import Qwen2VL_ViT_forward, Qwen2VLGeneration_forward
    from vidcom2_amd.models.qwen3_vl import Qwen3VLModel_forward

Same names, same installation (`types.MethodType(hook, model)`), same env knobs (`COMPRESSOR`,
`R_RATIO`); implemented as wrappers around the installed model's own methods, see `_intercept.py`.
"""
from __future__ import annotations

import types

from ._intercept import compressor_enabled

__all__ = ["install"]


def install(model, force: bool = False) -> bool:
    """Bind the matching hook on `model` the way the reference's lmms-eval wrappers do
    (lmms_eval/models/llava_onevision.py:157-163, README.md:76-93): only when `COMPRESSOR=vidcom2`
    (or `force`), with `types.MethodType`.  Returns whether a hook was bound.

    LLaVA-OneVision / LLaVA-Video : `model.prepare_inputs_labels_for_multimodal`
    Qwen2.5-VL / Qwen2-VL / Qwen3-VL : `forward` of the inner `*Model` (`model.model` of a
                                    `*ForConditionalGeneration`)
    """
    if not (force or compressor_enabled()):
        return False
    names = {k.__name__ for k in type(model).__mro__}
    if hasattr(model, "prepare_inputs_labels_for_multimodal"):
        from .llava import cus_prepare_inputs_labels_for_multimodal as hook
        model.__dict__["prepare_inputs_labels_for_multimodal"] = types.MethodType(hook, model)
        return True
    inner = model.model if any(n.endswith("ForConditionalGeneration") for n in names) else model
    inner_names = {k.__name__ for k in type(inner).__mro__}
    if any(n.startswith("Qwen3VL") for n in inner_names):
        from .qwen3_vl import Qwen3VLModel_forward as hook
    elif any(n.startswith("Qwen2_5_VL") for n in inner_names):
        from .qwen2_5_vl import Qwen2_5_VLModel_forward as hook
    elif any(n.startswith("Qwen2VL") for n in inner_names):
        from .qwen2_vl import Qwen2VLModel_forward as hook
    else:
        raise TypeError(f"vidcom2_amd.models.install: no hook for {type(model).__name__}")
    inner.__dict__["forward"] = types.MethodType(hook, inner)
    return True
