"""vidcom2_amd -- MI355X-native VidCom2 token compression (drop-in for token_compressor.vidcom2).

Same names as the reference package (token_compressor/vidcom2/__init__.py:2-22); the tensor
work runs in hand-written gfx950 HIP kernels behind the C ABI in include/vc2.h.
"""
from .vidcom2 import (  # noqa: F401
    MODEL_SPECS,
    vidcom2_compression,
    select_low_var_channels,
    compute_gaussian_scores,
    _multi_scale_gaussian,
    compute_scales,
    select_outlier_indices,
    map_features,
    _map_linear_offset,
    _map_grid_vid,
    compress,
    compress_batch,
    CompressionResult,
)

__all__ = [
    "vidcom2_compression",
    "select_low_var_channels",
    "compute_gaussian_scores",
    "compute_scales",
    "select_outlier_indices",
    "map_features",
    "_map_linear_offset",
    "_map_grid_vid",
]
