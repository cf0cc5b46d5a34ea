"""Bit-portable synthetic [F, N, D] frame-token embeddings.

Every value is produced by integer arithmetic (splitmix64) followed by ONE
int->fp32 conversion and fp32 multiplies/adds evaluated one numpy ufunc at a
time (no FMA contraction), then one round-to-nearest-even cast to the target
dtype. The result is therefore bit-identical on any IEEE-754 machine, which is
what lets the golden fixtures under tests/golden/ store only a seed plus the
sha256 of the generated tensor instead of the tensor itself.

Distributions
  "iid"   x = g                       (plain ~N(0,1))
  "drift" x[f,n,c] = s_c * ((b[n,c] + 0.3*e[f,n,c]) + t_f * d[c]),
          t_f = f/(F-1): a static per-position "scene" b shared by all frames,
          per-frame noise e, and a slow global drift d so later frames move away
          from the video centre -> non-trivial per-frame budgets.  s_c is a
          per-channel scale in {2^(k/2)} so channel variances spread over ~2 decades.
  "cancel" adversarial for the centre means: large positive, tiny, large negative thirds per frame and channel, so the
          tiny addends are rounded against a large running sum that later cancels (see make_fp32_frames).
"""
from __future__ import annotations

import hashlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_INV_STD = np.float32(1.0 / 37837.2)   # Irwin-Hall(4) of 16-bit uniforms has std 37837.2
_SQRT2 = np.float32(1.41421356)

_TID_B, _TID_E, _TID_D, _TID_S, _TID_IID = 1, 2, 3, 4, 5


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def gauss(seed: int, tensor_id: int, start: int, count: int) -> np.ndarray:
    """count ~N(0,1) fp32 variates for flat indices [start, start+count) of stream (seed, tensor_id)."""
    idx = np.arange(start, start + count, dtype=np.uint64)
    # stream base = splitmix64(seed * 2^8 + tensor_id): decorrelates neighbouring seeds/streams
    base = _splitmix64(np.array([(int(seed) << 8) + int(tensor_id)], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        key = base + idx
    h = _splitmix64(key)
    m = np.uint64(0xFFFF)
    s = (h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48))
    return (s.astype(np.int64) - 131070).astype(np.float32) * _INV_STD


def _channel_scale(seed: int, D: int) -> np.ndarray:
    g = gauss(seed, _TID_S, 0, D)
    k = np.rint(g * np.float32(2.0)).astype(np.int64)          # half-octave steps
    k = np.clip(k, -6, 6)
    base = np.ldexp(np.float32(1.0), (k >> 1).astype(np.int32)).astype(np.float32)
    return np.where((k & 1) == 1, base * _SQRT2, base).astype(np.float32)


def make_fp32_frames(F: int, N: int, D: int, f0: int, count: int, seed: int = 0, dist: str = "drift") -> np.ndarray:
    """fp32 frames [f0, f0+count) of the F-frame video (lets each rank build only its own shard)."""
    out = np.empty((count, N, D), dtype=np.float32)
    # frames are consecutive flat indices of their streams: small frames are generated several at a time (one numpy call over
    # ~1M elements instead of one per 50k-element frame; every element is the same function of its index: same bits)
    grp = max(1, (1 << 20) // max(1, N * D))
    if dist == "iid":
        for i in range(0, count, grp):
            g = min(grp, count - i)
            out[i:i + g] = gauss(seed, _TID_IID, (f0 + i) * N * D, g * N * D).reshape(g, N, D)
        return out
    if dist == "cancel":
        # ADVERSARIAL for the centre means (vidcom2.py:51-52).  fp32 sums of T values are EXACT while the addends are
        # within 16 (bf16) / 13 (fp16) binades of the running sum -- which is why torch's fp32 cascade and the exact
        # sum agree to a few ulps on ordinary data -- so: every channel sees a third of a frame's tokens at +a, then a
        # third with values five orders of magnitude smaller, then a third at -a,
        #     x[f,n,c] = a_c * (z(n, c) + 0.02 e[f,n,c]),  z = +1 / 1e-5 e' / -1 by ((n + 64 (c % 3)) % 192) // 64,
        # a_c = 1 (c < D/2) or 1.5 (the rest: higher variance, not scored); the channel phase c % 3 keeps two thirds
        # of every ROW large, so the tiny values survive the normalisation.  The small addends meet a running sum of
        # half of sum |x^| and the big ones then cancel: the cascade's error is hundreds of ulps OF THE MEAN.
        n_idx = np.arange(N).reshape(N, 1)
        c_idx = np.arange(D).reshape(1, D)
        zone = ((n_idx + 64 * (c_idx % 3)) % 192) // 64
        zone = np.where(n_idx >= (N // 192) * 192, 1, zone)              # leftover tokens: the tiny zone
        big = np.where(zone == 0, np.float32(1.0), np.where(zone == 2, np.float32(-1.0), np.float32(0.0))).astype(np.float32)
        tiny = (zone == 1)
        amp = np.where(np.arange(D) < D // 2, np.float32(1.0), np.float32(1.5)).astype(np.float32)
        for i in range(0, count, grp):
            g = min(grp, count - i)
            e = gauss(seed, _TID_E, (f0 + i) * N * D, g * N * D).reshape(g, N, D)
            v = np.where(tiny, np.float32(1e-5) * e, big + np.float32(0.02) * e).astype(np.float32)
            out[i:i + g] = amp * v
        return out
    if dist != "drift":
        raise ValueError(f"unknown dist {dist!r}")
    b = gauss(seed, _TID_B, 0, N * D).reshape(N, D)
    d = gauss(seed, _TID_D, 0, D)
    s = _channel_scale(seed, D)
    denom = np.float32(max(F - 1, 1))
    for i in range(0, count, grp):
        g = min(grp, count - i)
        e = gauss(seed, _TID_E, (f0 + i) * N * D, g * N * D).reshape(g, N, D)
        t = (np.arange(f0 + i, f0 + i + g, dtype=np.float32) / denom).reshape(g, 1, 1)
        v = b + np.float32(0.3) * e
        v = v + t * d
        out[i:i + g] = s * v
    return out


def make_fp32(F: int, N: int, D: int, seed: int = 0, dist: str = "drift") -> np.ndarray:
    """fp32 [F, N, D] master tensor (cast with :func:`to_torch`)."""
    return make_fp32_frames(F, N, D, 0, F, seed, dist)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit patterns (uint16), NaN-preserving."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    lsb = (u >> np.uint32(16)) & np.uint32(1)
    r = ((u + np.uint32(0x7FFF) + lsb) >> np.uint32(16)).astype(np.uint16)
    nan = (u & np.uint32(0x7FFFFFFF)) > np.uint32(0x7F800000)
    return np.where(nan, ((u >> np.uint32(16)) | np.uint32(0x40)).astype(np.uint16), r)


def to_torch(x32: np.ndarray, dtype):
    """Cast the fp32 master to a torch tensor of `dtype` (fp32 / bf16 / fp16), RNE."""
    import torch
    if dtype == torch.float32:
        return torch.from_numpy(np.ascontiguousarray(x32))
    if dtype == torch.bfloat16:                    # (cache-sized slices: the whole-array form allocates ten arrays of x32's size)
        bits = np.empty(x32.shape, dtype=np.uint16)
        _cast_into(bits, np.ascontiguousarray(x32, dtype=np.float32), "bf16")
        return torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16).reshape(x32.shape)
    if dtype == torch.float16:
        return torch.from_numpy(x32.astype(np.float16))
    raise ValueError(f"unsupported dtype {dtype}")


def _cast_into(dst: np.ndarray, v32: np.ndarray, kind: str) -> None:
    """RNE-cast fp32 v32 into dst (uint16 bf16 bit patterns / float16 / float32), cache-sized slices at a time."""
    a, d = v32.reshape(-1), dst.reshape(-1)
    for i in range(0, a.size, 1 << 18):
        j = min(a.size, i + (1 << 18))
        d[i:j] = f32_to_bf16_bits(a[i:j]) if kind == "bf16" else a[i:j]       # (float16 / float32: numpy casts RNE)


def make(F: int, N: int, D: int, dtype, seed: int = 0, dist: str = "drift"):
    """[F*N, D] torch tensor in `dtype` -- the `flattened_feat` the hooks hand to the path.  Large tensors are
    produced a few frames at a time by a handful of threads (frames are independent element-wise functions of the
    seed and the index: same bits as make_fp32 + to_torch, at a fraction of the time and memory)."""
    import torch
    if F * N * D < (1 << 23) or F < 8:
        return to_torch(make_fp32(F, N, D, seed, dist), dtype).reshape(F * N, D)
    import os
    from concurrent.futures import ThreadPoolExecutor
    if dtype == torch.bfloat16:
        kind, out = "bf16", np.empty((F, N, D), dtype=np.uint16)
    elif dtype == torch.float16:
        kind, out = "f16", np.empty((F, N, D), dtype=np.float16)
    elif dtype == torch.float32:
        kind, out = "f32", np.empty((F, N, D), dtype=np.float32)
    else:
        raise ValueError(f"unsupported dtype {dtype}")
    step = max(4, min(64, (1 << 22) // (N * D)))              # frames per piece (>= 4: the per-call setup of `drift`)

    def fill(f0):
        cnt = min(step, F - f0)
        _cast_into(out[f0:f0 + cnt], make_fp32_frames(F, N, D, f0, cnt, seed, dist), kind)

    with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
        list(ex.map(fill, range(0, F, step)))
    t = torch.from_numpy(out.view(np.int16)).view(torch.bfloat16) if kind == "bf16" else torch.from_numpy(out)
    return t.reshape(F * N, D)


def sha256_tensor(t) -> str:
    """sha256 over the raw bytes of a contiguous CPU tensor (dtype-agnostic)."""
    import torch
    t = t.detach().cpu().contiguous()
    if not t.numel():
        return hashlib.sha256(b"").hexdigest()
    return hashlib.sha256(memoryview(t.reshape(-1).view(torch.uint8).numpy())).hexdigest()      # (no copy of the bytes)
