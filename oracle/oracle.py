"""ctypes front-end of oracle/vc2_oracle.cpp (TEST INFRASTRUCTURE ONLY).

Each function mirrors one reference function (token_compressor/vidcom2/vidcom2.py, cited in
the C++ file) and takes/returns CPU torch tensors in the same dtype the reference would.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libvc2oracle.so")
_lib = None

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

__all__ = [
    "build", "chan_var", "topk_smallest", "select_low_var_channel_idx", "select_low_var_channels",
    "compute_gaussian_scores", "gaussian_debug", "fuse", "compute_scales", "compute_ks",
    "select_outlier_indices", "map_linear_offset", "map_grid_vid", "compress_indices",
    "vidcom2_compression", "set_num_threads", "exp_T", "gaussian_scores_sharded", "set_mode",
    "multi_scale_gaussian",
]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vc2_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


def _L():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_mode(mode: str) -> None:
    """'exact' (default): reductions accumulated exactly; 'torch': replay torch's CPU fp32 accumulation
    order for the L2 norm and the row sums (bit-exact to the reference in half precision)."""
    _L().vc2o_set_mode({"exact": 0, "torch": 1}[mode])


def set_num_threads(n: int) -> None:
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(n))
    except OSError:
        pass


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _i64(v):
    return ctypes.c_int64(int(v))


def _chk(rc, what):
    if rc == -2:
        raise RuntimeError(f"{what}: shape is invalid for input (rows not divisible by tokens-per-frame)")
    if rc == -3:
        raise RuntimeError(f"{what}: selected index k out of range")
    if rc != 0:
        raise RuntimeError(f"{what}: oracle error {rc}")


def _prep(x):
    assert x.device.type == "cpu" and x.dtype in _DT, (x.device, x.dtype)
    return x.contiguous()


def chan_var(x: torch.Tensor) -> torch.Tensor:
    x = _prep(x)
    R, D = x.shape
    out = torch.empty(D, dtype=x.dtype)
    _chk(_L().vc2o_chan_var(_p(x), _i64(R), _i64(D), _DT[x.dtype], _p(out)), "chan_var")
    return out


def topk_smallest(v: torch.Tensor, k: int, sorted: bool = True) -> torch.Tensor:
    v = _prep(v)
    out = torch.empty(k, dtype=torch.int64)
    _chk(_L().vc2o_topk_smallest(_p(v), _i64(v.numel()), _i64(k), int(sorted), _DT[v.dtype], _p(out)), "topk")
    return out


def select_low_var_channel_idx(x: torch.Tensor, ratio: float = 0.5) -> Tuple[torch.Tensor, torch.Tensor]:
    x = _prep(x)
    R, D = x.shape
    k = int(D * ratio)
    var = torch.empty(D, dtype=x.dtype)
    idx = torch.empty(k, dtype=torch.int64)
    _chk(_L().vc2o_select_low_var_channels(_p(x), _i64(R), _i64(D), _DT[x.dtype], _i64(k), _p(var), _p(idx)),
         "select_low_var_channels")
    return idx, var


def select_low_var_channels(x: torch.Tensor, ratio: float = 0.5) -> torch.Tensor:
    idx, _ = select_low_var_channel_idx(x, ratio)
    return x[:, idx]


def gaussian_debug(x: torch.Tensor, chan_idx: torch.Tensor, tpf: int) -> dict:
    """Scores plus the intermediates (norm, centres, squared distances)."""
    x = _prep(x)
    R, D = x.shape
    C = chan_idx.numel()
    if tpf <= 0 or R % tpf:
        _chk(-2, "compute_gaussian_scores")
    F = R // tpf
    T = x.dtype
    o = dict(v=torch.empty(F, tpf, dtype=T), f=torch.empty(F, tpf, dtype=T), norm=torch.empty(R, dtype=T),
             vid_center=torch.empty(C, dtype=T), frame_center=torch.empty(F, C, dtype=T),
             dist_v=torch.empty(F, tpf, dtype=T), dist_f=torch.empty(F, tpf, dtype=T))
    ci = chan_idx.contiguous().to(torch.int64)
    _chk(_L().vc2o_gaussian_scores(_p(x), _i64(R), _i64(D), _DT[T], _p(ci), _i64(C), _i64(tpf), _p(o["v"]),
                                   _p(o["f"]), _p(o["norm"]), _p(o["vid_center"]), _p(o["frame_center"]),
                                   _p(o["dist_v"]), _p(o["dist_f"])), "compute_gaussian_scores")
    return o


def compute_gaussian_scores(sel: torch.Tensor, tpf: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference signature: takes the channel-selected features [F*N, C']."""
    sel = _prep(sel)
    o = gaussian_debug(sel, torch.arange(sel.shape[1]), tpf)
    return o["v"], o["f"]


def multi_scale_gaussian(x: torch.Tensor, center: torch.Tensor, alphas) -> torch.Tensor:
    """vidcom2.py:59-62 standalone.  x: [F, N, C]; center: [1, 1, C] or [F, 1, C]; returns [F, N]."""
    if x.dim() != 3 or center.dim() != 3 or center.shape[1] != 1 or center.shape[2] != x.shape[2] \
            or center.shape[0] not in (1, x.shape[0]):
        raise RuntimeError("multi_scale_gaussian: x [F,N,C] and center [1|F,1,C] expected")
    F, N, C = x.shape
    xx, cc = _prep(x.reshape(F * N, C)), _prep(center.reshape(-1, C).to(x.dtype))
    two_a = (ctypes.c_float * len(alphas))(*[float(2 * a) for a in alphas])
    out = torch.empty(F, N, dtype=x.dtype)
    _chk(_L().vc2o_multi_scale_gaussian(_p(xx), _i64(F * N), _i64(C), _DT[x.dtype], _p(cc), _i64(cc.shape[0]),
                                        _i64(N), two_a, ctypes.c_int(len(alphas)), _p(out)), "multi_scale_gaussian")
    return out


def fuse(v: torch.Tensor, f: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    v, f = _prep(v), _prep(f)
    F, N = v.shape
    s = torch.empty(F, dtype=v.dtype)
    tot = torch.empty(F, N, dtype=v.dtype)
    _chk(_L().vc2o_fuse(_p(v), _p(f), _i64(F), _i64(N), _DT[v.dtype], _p(s), _p(tot)), "fuse")
    return s, tot


def compute_scales(scores: torch.Tensor, base: float, temp: float = 0.01) -> torch.Tensor:
    scores = _prep(scores)
    out = torch.empty_like(scores)
    _chk(_L().vc2o_compute_scales(_p(scores), _i64(scores.numel()), ctypes.c_double(base), ctypes.c_double(temp),
                                  _DT[scores.dtype], _p(out)), "compute_scales")
    return out


def compute_ks(scales: torch.Tensor, tpf: int) -> List[int]:
    scales = _prep(scales)
    ks = torch.empty(scales.numel(), dtype=torch.int64)
    _chk(_L().vc2o_ks(_p(scales), _i64(scales.numel()), _i64(tpf), _DT[scales.dtype], _p(ks)), "ks")
    return ks.tolist()


def select_outlier_indices(scores: torch.Tensor, scales: torch.Tensor, tpf: int) -> List[torch.Tensor]:
    scores = _prep(scores)
    F, N = scores.shape
    ks = torch.tensor(compute_ks(scales, tpf), dtype=torch.int64)
    out = torch.empty(int(ks.clamp(max=N).sum()) if F else 0, dtype=torch.int64)
    _chk(_L().vc2o_select_outliers(_p(scores), _i64(F), _i64(N), _DT[scores.dtype], _p(ks), _p(out)),
         "select_outlier_indices")
    return list(torch.split(out, ks.tolist()))


def map_linear_offset(indices: List[torch.Tensor], tpf: int) -> torch.Tensor:
    ks = torch.tensor([i.numel() for i in indices], dtype=torch.int64)
    loc = torch.cat(indices).contiguous()
    out = torch.empty_like(loc)
    _L().vc2o_map_linear(_p(loc), _p(ks), _i64(len(indices)), _i64(tpf), _p(out))
    return out


def map_grid_vid(indices: List[torch.Tensor], h: int) -> torch.Tensor:
    ks = torch.tensor([i.numel() for i in indices], dtype=torch.int64)
    loc = torch.cat(indices).contiguous()
    out = torch.empty(loc.numel() + len(indices) * h, dtype=torch.int64)
    _L().vc2o_map_grid_vid(_p(loc), _p(ks), _i64(len(indices)), _i64(h), _p(out))
    return out


def gaussian_scores_sharded(x: torch.Tensor, chan_idx: torch.Tensor, tpf: int, csum_all=None, R_total: int = 0):
    """Shard-local scores.  Returns (v, f, csum_local[C] float64); csum_all [P, C] float64 (all shards'
    csum_local stacked) + R_total switch the video centre to the whole video's."""
    x = _prep(x)
    R, D = x.shape
    C = chan_idx.numel()
    if tpf <= 0 or R % tpf:
        _chk(-2, "compute_gaussian_scores")
    F, T = R // tpf, x.dtype
    v, f = torch.empty(F, tpf, dtype=T), torch.empty(F, tpf, dtype=T)
    csum = torch.empty(C, dtype=torch.float64)
    ci = chan_idx.contiguous().to(torch.int64)
    cin = csum_all.contiguous().to(torch.float64) if csum_all is not None else None
    P = cin.shape[0] if cin is not None else 0
    _chk(_L().vc2o_gaussian_scores_ex(_p(x), _i64(R), _i64(D), _DT[T], _p(ci), _i64(C), _i64(tpf), _p(cin), _i64(P),
                                      _i64(R_total), _p(csum), _p(v), _p(f), _p(None), _p(None), _p(None), _p(None),
                                      _p(None)), "gaussian_scores_sharded")
    return v, f, csum


def exp_T(x: torch.Tensor) -> torch.Tensor:
    """RN_T(exp(x)) as the Gaussian kernel evaluates it."""
    x = _prep(x)
    out = torch.empty_like(x)
    _L().vc2o_exp_T(_p(x), _i64(x.numel()), _DT[x.dtype], _p(out))
    return out


def compress_indices(x: torch.Tensor, tpf: int, base: float = 0.25) -> dict:
    """Whole pass (channel select -> scores -> budgets -> per-frame selection -> linear map)."""
    x = _prep(x)
    R, D = x.shape
    if tpf <= 0 or R % tpf:
        _chk(-2, "compress_indices")
    F, T = R // tpf, x.dtype
    o = dict(chan_idx=torch.empty(int(D * 0.5), dtype=torch.int64), v=torch.empty(F, tpf, dtype=T),
             f=torch.empty(F, tpf, dtype=T), total=torch.empty(F, tpf, dtype=T), s=torch.empty(F, dtype=T),
             scales=torch.empty(F, dtype=T), ks=torch.empty(F, dtype=torch.int64))
    gidx = torch.empty(R, dtype=torch.int64)
    K = ctypes.c_int64(0)
    _chk(_L().vc2o_compress_indices(_p(x), _i64(F), _i64(tpf), _i64(D), _DT[T], ctypes.c_double(base),
                                    _p(o["chan_idx"]), _p(o["v"]), _p(o["f"]), _p(o["total"]), _p(o["s"]),
                                    _p(o["scales"]), _p(o["ks"]), _p(gidx), ctypes.byref(K)), "compress_indices")
    o["global_idx"] = gidx[: K.value].clone()
    return o


_SPECS = {"llava_ov": (196, "linear"), "llava_vid": (169, "grid_vid"), "qwen2_vl": (None, "linear"),
          "qwen2_5_vl": (None, "linear"), "qwen3_vl": (None, "linear")}


def vidcom2_compression(flat: torch.Tensor, model: str = "llava_ov", base_scale: float = 0.25,
                        frame_token_len=None, img_feat=None) -> torch.Tensor:
    """vidcom2.py:15-36 end to end (CPU oracle)."""
    if model not in _SPECS:
        raise ValueError(f"Unknown model: {model}")
    tpf, mapper = _SPECS[model]
    if tpf is None:
        tpf = frame_token_len
    if tpf is None:
        raise ValueError(f"frame_token_len required for {model}")
    tpf = int(tpf)
    o = compress_indices(flat, tpf, base_scale)
    if mapper == "linear":
        return flat[o["global_idx"]]
    if img_feat is None:
        raise ValueError("img_feat required for grid mapping")
    F = flat.shape[0] // tpf
    loc = o["global_idx"] - torch.repeat_interleave(torch.arange(F) * tpf, o["ks"])
    return img_feat[map_grid_vid(list(torch.split(loc, o["ks"].tolist())), 13)]
