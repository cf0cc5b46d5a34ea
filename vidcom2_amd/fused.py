"""Hook-side fusion: kept rows written once, directly at their final positions.

Host mirror of `vc2_gather_scatter` / `vc2_keep_positions` (include/vc2.h).  They replace, in the hooks,

* `flat[global_idx]` followed by `torch.cat((rows, image_newline[None]))` (reference vidcom2.py:91 and
  models/llava.py:160-168): one launch writes the K kept rows and the newline row behind them
  (`CompressPlan(..., tail_rows=1)` does this inside the pass itself);
* `inputs_embeds[:, keep_token_indices, :]` (models/qwen2_5_vl.py:153-182): `keep_positions` builds the kept
  sequence positions on the device (no `nonzero` round trips), `gather_scatter` copies text and kept video rows in
  one launch;
* the N+1 gathers of Qwen3-VL's deep-stack (models/qwen3_vl.py:140-165): several tensors, one index list, one launch;
* (f3) LLaVA's `get_2dPool` (llava/model/llava_arch.py:171-190) followed by the pass's first sweep: `pool_stats`
  writes the pooled video once and leaves its channel statistics in the pass's workspace.

Device tensors only: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from ._ffi import DTYPE_CODE, check, lib, on_device, ptr, require_device, stream_ptr

_check_rc = check          # (gather_scatter has a keyword argument named `check`)

__all__ = ["gather_scatter", "keep_positions", "pool_stats", "POOL_MODES"]

MAX_SOURCES = 8


def _ptr_array(ts: Sequence[torch.Tensor]):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _i64_array(vs: Sequence[int]):
    return (ctypes.c_int64 * len(vs))(*[int(v) for v in vs])


def gather_scatter(srcs: Sequence[torch.Tensor], idx: Optional[torch.Tensor] = None, n: Optional[int] = None,
                   n_dev: Optional[torch.Tensor] = None, dst_pos: Optional[torch.Tensor] = None,
                   dsts: Optional[Sequence[torch.Tensor]] = None, dst_row0: int = 0,
                   tail: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None,
                   check: bool = False) -> List[torch.Tensor]:
    """dsts[t][dst_pos[j] or dst_row0 + j] = srcs[t][idx[j] or j] for j < n, every tensor in ONE launch; `tail`
    rows ([m, D]) are appended to dsts[0] behind the gathered ones.

    srcs: 2-D tensors [rows_t, D] of one dtype / feature size.  n defaults to len(idx) (or rows of srcs[0]); n_dev
    (device int64[1]) caps it on the device.  dsts default to fresh [n + m, D] (first) / [n, D] tensors.
    A row index outside its tensor is skipped on the device and reported: in `status` (device int32[1], bit 1) when
    the caller passes one, and -- check=True -- by an IndexError after one small read-back, like
    the torch indexing this replaces (an unreported skip would leave an uninitialised row in a fresh destination).  The
    model hooks pass lists that keep_positions has just validated, so they skip the second read-back.
    """
    srcs = list(srcs)
    if not 1 <= len(srcs) <= MAX_SOURCES:
        raise ValueError(f"gather_scatter takes 1..{MAX_SOURCES} tensors, got {len(srcs)}")
    x0 = srcs[0]
    require_device(x0, "srcs[0]")
    if x0.dtype not in DTYPE_CODE:
        raise TypeError(f"unsupported dtype {x0.dtype} (fp32 / bf16 / fp16 only)")
    D = x0.shape[-1]
    for i, s in enumerate(srcs):
        if s.dim() != 2 or s.shape[1] != D or s.dtype != x0.dtype or s.device != x0.device:
            raise RuntimeError(f"srcs[{i}]: expected a [rows, {D}] {x0.dtype} tensor on {x0.device}, got "
                               f"{tuple(s.shape)} {s.dtype} on {s.device}")
    srcs = [s if s.is_contiguous() else s.contiguous() for s in srcs]
    if idx is not None:
        if idx.dtype != torch.int64 or idx.device != x0.device or idx.dim() != 1:
            raise RuntimeError("idx must be a 1-D int64 tensor on the tensors' device")
        idx = idx.contiguous()
    if n is None:
        n = int(idx.numel()) if idx is not None else int(x0.shape[0])
    n = int(n)
    if idx is not None and n > idx.numel():
        raise IndexError(f"n={n} exceeds len(idx)={idx.numel()}")
    if dst_pos is not None:
        if dst_pos.dtype != torch.int64 or dst_pos.device != x0.device or dst_pos.numel() < n:
            raise RuntimeError("dst_pos must be an int64 tensor of at least n entries on the tensors' device")
        dst_pos = dst_pos.contiguous()
    m = 0
    if tail is not None:
        if tail.dim() == 1:
            tail = tail[None]
        if tail.dim() != 2 or tail.shape[1] != D or tail.dtype != x0.dtype or tail.device != x0.device:
            raise RuntimeError(f"tail must be [m, {D}] {x0.dtype} on {x0.device}")
        tail = tail.contiguous()
        m = int(tail.shape[0])
    if dsts is None:
        if dst_pos is not None:
            raise ValueError("dst_pos needs caller-provided dsts")
        dsts = [torch.empty((dst_row0 + n + (m if t == 0 else 0), D), dtype=x0.dtype, device=x0.device)
                for t in range(len(srcs))]
    else:
        dsts = list(dsts)
        if len(dsts) != len(srcs):
            raise ValueError("one destination per source")
        for i, d in enumerate(dsts):
            if d.dim() != 2 or d.shape[1] != D or d.dtype != x0.dtype or d.device != x0.device or not d.is_contiguous():
                raise RuntimeError(f"dsts[{i}] must be a contiguous [rows, {D}] {x0.dtype} tensor on {x0.device}")
    if check and status is None:
        status = torch.zeros(1, dtype=torch.int32, device=x0.device)
    with on_device(x0.device):
        rc = lib().vc2_gather_scatter(_ptr_array(srcs), _i64_array([s.shape[0] for s in srcs]), _ptr_array(dsts),
                                      _i64_array([d.shape[0] for d in dsts]), len(srcs), D, DTYPE_CODE[x0.dtype],
                                      ptr(idx), ptr(n_dev), n, ptr(dst_pos), int(dst_row0), ptr(tail), m, ptr(status),
                                      stream_ptr(x0.device))
    _check_rc(rc, "vc2_gather_scatter")
    if check and int(status.item()) & 2:
        raise IndexError("gather_scatter: a row index is out of range for its source or destination tensor")
    return dsts


_PINNED = threading.local()


def _pinned_counts() -> torch.Tensor:
    """Four pinned int64 words per thread for vc2_keep_positions' counts (used synchronously: the call spins until the
    kernel has written them; never freed -- see vidcom2._KHost)."""
    t = getattr(_PINNED, "counts", None)
    if t is None:
        t = _PINNED.counts = torch.zeros(4, dtype=torch.int64).pin_memory()
    return t


def keep_positions(video_mask: torch.Tensor, kept: torch.Tensor, n_video: Optional[int] = None,
                   visual_mask: Optional[torch.Tensor] = None,
                   n_visual: Optional[int] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Sorted sequence positions a pruned prefill keeps: every non-video position plus the video tokens whose
    ordinal is in `kept` (sorted int64).  With `visual_mask` also the rows the deep-stack tensors keep (ordinals among
    the visual positions).  n_video / n_visual = number of True entries of video_mask / visual_mask (known to the
    hooks: the row counts of the video embeds / deep-stack tensors); each one omitted costs a sync."""
    vm = video_mask.reshape(-1)
    vm = vm.view(torch.uint8) if vm.dtype == torch.bool else vm.to(torch.uint8)
    vm = vm.contiguous()
    require_device(vm, "video_mask")
    S = int(vm.numel())
    kept = kept.to(device=vm.device, dtype=torch.int64).contiguous()
    K = int(kept.numel())
    if n_video is None:
        n_video = int(vm.sum().item())
    n_keep = S - int(n_video) + K
    if n_keep < 0 or int(n_video) < K:
        raise IndexError(f"keep_positions: {K} kept ordinals for {n_video} video positions among {S}")
    keep = torch.empty(n_keep, dtype=torch.int64, device=vm.device)
    vis = vis_rows = None
    if visual_mask is not None:
        vis = visual_mask.reshape(-1)
        vis = (vis.view(torch.uint8) if vis.dtype == torch.bool else vis.to(torch.uint8)).contiguous()
        if vis.numel() != S or vis.device != vm.device:
            raise RuntimeError("visual_mask must cover the same positions as video_mask")
        vis_rows = torch.empty(S, dtype=torch.int64, device=vm.device)
    # The counts / error word come back through pinned host memory the kernel stores to (no device-to-host copy: the
    # host spins on the error word, which the kernel writes last) -- what the index lists are worth: torch indexing,
    # which this replaces, raises on a bad index; an unchecked list would gather rows from wherever it points.
    counts = _pinned_counts()
    counts[2] = -1
    with on_device(vm.device):
        rc = lib().vc2_keep_positions(ptr(vm), S, ptr(kept), None, K, ptr(vis), ptr(keep), n_keep, ptr(vis_rows),
                                      S if vis_rows is not None else 0, ctypes.c_void_p(counts.data_ptr()),
                                      stream_ptr(vm.device))
    check(rc, "vc2_keep_positions")
    if int(lib().vc2_wait_host_count(ctypes.c_void_p(counts.data_ptr() + 16), 0.05)) < 0:
        torch.cuda.current_stream(vm.device).synchronize()       # (a busy stream: the plain wait)
    found, found_vis, err = (int(v) for v in counts[:3].tolist())
    if err:
        why = []
        if err & 1 or err & 8:
            why.append(f"video_mask holds {S - (found - K)} video positions, the caller said {n_video}")
        if err & 4:
            why.append("kept must be strictly ascending ordinals of video positions")
        if err & 2:
            why.append("more visual rows than positions")
        raise IndexError("keep_positions: " + "; ".join(why))
    if vis_rows is not None:
        # video tokens are visual tokens: the kept visual rows are the non-video visual ones plus the K kept
        if n_visual is not None and int(n_visual) - int(n_video) + K != found_vis:
            raise IndexError(f"keep_positions: visual_mask flags {found_vis + int(n_video) - K} positions, the caller said {n_visual}")
        vis_rows = vis_rows[:found_vis]
    return keep, vis_rows


POOL_MODES = {"average": 1, "max": 2, "bilinear": 3}


def pool_stats(image_feature: torch.Tensor, height: int, width: int, mode: str = "average"):
    """2x2 pool of the projector output `[F, height*width, D]` (get_2dPool with stride 2, in the token-major layout it
    permutes from and back to) fused with sweep 1 of the compression pass.

    Returns (pooled `[F, h*w, D]`, workspace): hand the workspace to `compress(..., stats_ws=workspace)` /
    `CompressPlan(..., ws=workspace).enqueue(..., have_stats=True)` on the same stream and the pass skips its first
    sweep.  All three modes reproduce torch's bits (avg_pool2d / max_pool2d / interpolate(mode="bilinear") on the x86
    CPU build; bilinear needs D % 8 == 0 in fp32, D % 16 == 0 in 16-bit, see include/vc2.h)."""
    from . import _ffi
    require_device(image_feature, "image_feature")
    if mode not in POOL_MODES:
        raise ValueError(f"Unexpected mm_spatial_pool_mode: {mode}")
    x = image_feature
    if x.dim() != 3 or x.shape[1] != int(height) * int(width):
        raise RuntimeError(f"image_feature must be [frames, {height}*{width}, dim], got {tuple(x.shape)}")
    if x.dtype not in DTYPE_CODE:
        raise TypeError(f"unsupported dtype {x.dtype} (fp32 / bf16 / fp16 only)")
    x = x if x.is_contiguous() else x.contiguous()
    F, _, D = x.shape
    h = ctypes.c_int64(0)
    w = ctypes.c_int64(0)
    check(lib().vc2_pool_out_tokens(int(height), int(width), POOL_MODES[mode], ctypes.byref(h), ctypes.byref(w)),
          "vc2_pool_out_tokens")
    n = int(h.value) * int(w.value)
    ws = _ffi.workspace(F, n, D, x.dtype, x.device)
    out = torch.empty((F, n, D), dtype=x.dtype, device=x.device)
    with on_device(x.device):
        rc = lib().vc2_pool_stats(ptr(x), F, int(height), int(width), D, DTYPE_CODE[x.dtype], POOL_MODES[mode], ptr(ws),
                                  ws.numel(), ptr(out), stream_ptr(x.device))
    check(rc, "vc2_pool_stats")
    # the partials' layout depends on (frames, tokens per frame): the workspace says what it was written for, and
    # `compress` / `CompressPlan.enqueue(have_stats=True)` use the statistics only for exactly that tensor and shape
    ws._vc2_stats_for = (int(F), int(n), int(D), x.dtype, out.data_ptr())
    return out, ws
