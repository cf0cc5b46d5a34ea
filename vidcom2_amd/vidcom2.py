"""MI355X-native mirror of the reference plugin API ``token_compressor/vidcom2/vidcom2.py``.

Same function names, argument meaning, return types and error behaviour as the reference
(file:line cited per function); the tensor work is done by the gfx950 HIP kernels in
``csrc/vc2_kernels.hip`` through the C ABI of ``include/vc2.h``.  PyTorch is used only to
allocate device memory and to supply the current HIP stream.

Numerics: like the reference, every op runs "in the input dtype" -- see DESIGN.md
"Numerics contract".  Inputs are never modified; every return value is a fresh tensor on the
input's device.  Tensors must live on a ROCm device: there is no CPU fallback.
"""
from __future__ import annotations

import atexit
import collections
import ctypes
import os
import threading
import time
import warnings
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _ffi
from ._ffi import DTYPE_CODE, MAP_GRID_VID, MAP_LINEAR, MAP_LOCAL, check, lib, on_device, ptr, require_device, stream_ptr

# vidcom2.py:7-13 -- 'mapper' selects the index mapping, 'tpf' the default tokens per frame
# (None = provided per call through frame_token_len).
MODEL_SPECS: Dict[str, Dict[str, Any]] = {
    "llava_ov": {"tpf": 196, "mapper": "linear"},
    "llava_vid": {"tpf": 169, "mapper": "grid_vid", "grid": 13},
    "qwen2_vl": {"tpf": None, "mapper": "linear"},
    "qwen2_5_vl": {"tpf": None, "mapper": "linear"},
    "qwen3_vl": {"tpf": None, "mapper": "linear"},
}
_DYNAMIC_TPF = {"qwen2_vl", "qwen2_5_vl", "qwen3_vl"}
_ALPHAS = [2 ** k for k in range(-3, 2)]          # vidcom2.py:54


def _first_tensor(a):
    if isinstance(a, torch.Tensor):
        return a
    if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
        return a[0]
    return None


def _guarded(fn):
    """Run `fn` with the device of its first tensor argument current (see _ffi.on_device)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t = next((x for x in map(_first_tensor, args) if x is not None), None)
        if t is None or t.device.type != "cuda":
            return fn(*args, **kwargs)
        with on_device(t.device):
            return fn(*args, **kwargs)
    return wrapper


def _as_int(v) -> int:
    if isinstance(v, torch.Tensor):
        return int(v.reshape(-1)[0].item()) if v.numel() == 1 else int(v)
    return int(v)


def _prep(x: torch.Tensor, what: str) -> torch.Tensor:
    require_device(x, what)
    if x.dtype not in DTYPE_CODE:
        raise TypeError(f"{what}: unsupported dtype {x.dtype} (fp32 / bf16 / fp16 only)")
    return x if x.is_contiguous() else x.contiguous()


@dataclass
class CompressionResult:
    """Everything one pass produces (device tensors; ``K`` / ``ks`` are synced to host lazily)."""
    rows: Optional[torch.Tensor]        # [K, D] gathered rows (None when no gather was requested)
    global_idx: torch.Tensor            # int64 [K] mapped global indices
    ks: torch.Tensor                    # int64 [F] per-frame budgets
    K: int
    v_score: Optional[torch.Tensor] = None   # T [F, N]
    f_score: Optional[torch.Tensor] = None   # T [F, N]


def cascade_level_power(n: int) -> int:
    """torch's SumKernel.cpp level_power for a reduction over n elements: max(4, ceil_log2(n) // 4) -- 4 up to 2^19
    elements, 5 up to 2^23, 6 up to 2^27 (the block size 2^lp of the centre-mean replays, csrc cascade_lp)."""
    return max(4, (max(int(n), 1) - 1).bit_length() // 4)


def cascade_modelled(n: int) -> bool:
    """Is torch's summation order for a centre mean over n tokens replayed by the kernels (csrc cascade_modelled)?"""
    lp = cascade_level_power(n)
    return lp <= 6 and (((int(n) >> lp) + (1 << lp) - 1) >> lp) <= 8192


class CompressPlan:
    """Pre-allocated buffers for repeated passes over one (F, N, D, dtype) shape.

    ``enqueue`` launches the whole pass (8 kernels, no host round trip) on the current stream;
    ``finish`` performs the path's single device->host sync (the reference's ``.tolist()``,
    vidcom2.py:72) and slices the outputs to the K kept tokens.
    """

    def __init__(self, F: int, N: int, D: int, dtype, device, base_scale: float = 0.25,
                 mapper: str = "linear", grid_h: int = 0, want_scores: bool = False, gather: bool = True,
                 tail_rows: int = 0, ws: Optional[torch.Tensor] = None):
        if mapper not in ("linear", "grid_vid"):
            raise ValueError(f"unknown mapper {mapper!r}")
        self.F, self.N, self.D, self.dtype, self.device = int(F), int(N), int(D), dtype, torch.device(device)
        self.base_scale = float(base_scale)
        self.map_mode = MAP_LINEAR if mapper == "linear" else MAP_GRID_VID
        self.grid_h = int(grid_h)
        L = lib()
        cap = int(L.vc2_kept_capacity(self.F, self.N, self.base_scale))
        if self.map_mode == MAP_GRID_VID:
            cap += self.F * self.grid_h
        self.cap = cap
        if dtype != torch.float32 and _ffi.get_mode() == "torch" and not cascade_modelled(self.F * self.N):
            # beyond this size the centre-mean replays do not model torch's outer-sum cascade (SumKernel.cpp level_power
            # 4 / 5 / 6 are modelled, i.e. up to 2^25 tokens per video): say so instead of silently keeping the exactly
            # rounded means
            warnings.warn(f"vidcom2_amd: {self.F} x {self.N} tokens exceed the modelled range of torch's centre-mean "
                          "accumulation order (2^25 tokens per video); boundary-near centre values "
                          "keep their exactly rounded mean and may differ from the CPU reference by one ulp",
                          RuntimeWarning, stacklevel=3)
        self.ws = ws if ws is not None else _ffi.workspace(self.F, self.N, self.D, dtype, self.device)
        self.kout = torch.empty(2, dtype=torch.int64, device=self.device)    # both words written by every pass
        # tail_rows: extra rows (LLaVA's newline embedding) the gather launch appends behind the kept ones
        self.tail_rows = int(tail_rows) if gather else 0
        self._gather, self._want_scores = bool(gather), bool(want_scores)
        self._spare = None
        # (K, status) mirrored by the selection launch into pinned host memory (vc2_compress_ex2): finish(early=True) gets
        # the count from there -- ~15 us before the pass ends, no device-to-host copy -- and returns while the gather runs
        self._khost = None
        self._khost_armed = False
        self._late = None                                # a pass that returned on its early status words (see _settle)
        self.new_outputs()

    def _alloc_outputs(self):
        dt, dev = self.dtype, self.device
        return (torch.empty(self.cap, dtype=torch.int64, device=dev),
                torch.empty(self.F, dtype=torch.int64, device=dev),
                torch.empty((self.cap + self.tail_rows, self.D), dtype=dt, device=dev) if self._gather else None,
                torch.empty((self.F, self.N), dtype=dt, device=dev) if self._want_scores else None,
                torch.empty((self.F, self.N), dtype=dt, device=dev) if self._want_scores else None,
                torch.empty(2, dtype=torch.int64, device=dev))    # (K, status): both words written by every pass

    def new_outputs(self) -> None:
        """Fresh output tensors (kept rows, indices, budgets, scores) for the next pass.  A plan that is re-used for
        independent calls (the one-shot API's plan cache) calls this before every enqueue, so that a result handed out
        earlier is never overwritten -- the reference returns fresh tensors as well; workspace and status words stay.
        (`prepare_spare` may have made the set already, while the previous pass was running.)"""
        spare, self._spare = self._spare, None
        self.idx, self.ks, self.rows, self.v, self.f, self.kout = spare if spare is not None else self._alloc_outputs()

    def output_bytes(self) -> int:
        """Bytes of ONE output set (kept rows, indices, budgets, scores, status words)."""
        es = torch.empty(0, dtype=self.dtype).element_size()
        n = self.cap * 8 + self.F * 8 + 16
        if self._gather:
            n += (self.cap + self.tail_rows) * self.D * es
        if self._want_scores:
            n += 2 * self.F * self.N * es
        return n

    def cached_bytes(self) -> int:
        """What a cached plan keeps alive: workspace, the last output set, the spare one."""
        return int(self.ws.numel()) + self.output_bytes() * (2 if self._spare is not None else 1)

    def prepare_spare(self) -> None:
        """Allocate the NEXT call's output tensors now -- called by the one-shot API between enqueue and finish, i.e.
        while the GPU is busy with this pass -- so that the next call launches its first kernel ~25 us earlier (five
        allocations).  Still fresh tensors per call: nothing a caller holds is ever overwritten."""
        if self._spare is None:
            self._spare = self._alloc_outputs()

    def enqueue(self, flat: torch.Tensor, gather_src: Optional[torch.Tensor] = None,
                tail: Optional[torch.Tensor] = None, have_stats: bool = False, mirror: bool = False,
                stream: Optional["torch.cuda.Stream"] = None) -> None:
        """have_stats: `self.ws` already holds flat's sweep-1 partials (fused.pool_stats wrote them on this
        stream), so the pass starts at the variance reduction.  mirror: the count is also written to pinned host
        memory (finish(early=True))."""
        src = flat if gather_src is None else gather_src
        if have_stats:
            tag = getattr(self.ws, "_vc2_stats_for", None)
            if tag is not None and tag != (self.F, self.N, self.D, self.dtype, flat.data_ptr()):
                raise RuntimeError(f"have_stats: the workspace holds the statistics of a {tag[:4]} tensor, not of this "
                                   f"({self.F}, {self.N}, {self.D}, {self.dtype}) one")
        if self.tail_rows:
            if tail is None or tail.shape != (self.tail_rows, self.D) or tail.dtype != self.dtype \
                    or tail.device != self.device:
                raise RuntimeError(f"tail must be [{self.tail_rows}, {self.D}] {self.dtype} on {self.device}")
            tail = tail.contiguous()
        khost = None
        self._settle()                                   # (a previous pass that returned on its early status words: ANY next
        #                                                   enqueue of the plan reads its final word first, mirrored or not)
        if mirror:
            if self._khost is None:
                self._khost = _khost_get()               # (pinned words from the process-wide pool: never freed)
                self._khost_addr, self._khost_word, self._khost_final = self._khost.addr, self._khost.word, self._khost.final
            self._khost_word.value = -1                  # (host write: the previous pass of this plan was finished)
            self._khost_final.value = 0
            self._khost.in_flight = True
            khost = self._khost_addr
        self._khost_armed = bool(mirror)
        with on_device(self.device):
            rc = lib().vc2_compress_ex2(ptr(flat), self.F, self.N, self.D, DTYPE_CODE[self.dtype], self.base_scale,
                                        self.map_mode, self.grid_h, ptr(src if self.rows is not None else None),
                                        src.shape[0], ptr(self.ws), self.ws.numel(), ptr(self.rows), ptr(self.idx),
                                        self.cap, ptr(self.ks), ptr(self.kout), ptr(self.v), ptr(self.f),
                                        ptr(tail if self.tail_rows else None), self.tail_rows,
                                        1 if have_stats else 0, khost,
                                        stream_ptr(self.device) if stream is None else ctypes.c_void_p(stream.cuda_stream))
        if rc != 0 and mirror:                           # (nothing was launched: the mirror block is not in flight)
            self._khost.in_flight = False
            self._khost_armed = False
        check(rc, "vc2_compress")

    def settle_quietly(self) -> None:
        """The last word on a pass that returned on its early status words, for the places that must not raise -- the
        plan is dropped (`__del__`), evicted from or cleared out of the plan cache, or the process ends: a late
        selection-guard hit is REPORTED there (RuntimeWarning; a "cannot happen" bit -- the caller's result is already
        in its hands), where `_settle` raises it in front of the plan's next pass."""
        try:
            self._settle()
        except RuntimeError as e:
            warnings.warn(f"vidcom2_amd: {e}", RuntimeWarning, stacklevel=2)
        except Exception:                                # (interpreter shutdown: torch half gone)
            pass

    def __del__(self):
        if getattr(self, "_late", None) is not None:
            self.settle_quietly()
        kh = getattr(self, "_khost", None)
        if kh is not None:
            try:
                _khost_put(kh)
            except Exception:                            # (interpreter shutdown)
                pass

    def finish(self) -> CompressionResult:
        # (spinning on an event behind a copy to pinned memory instead of this blocking copy: measured, no gain -- torch's
        #  blocking copy already spins)
        if self._khost_armed:
            # the count from the pinned mirror: the caller goes on (slicing, the next launches) while the gather launch is
            # still running -- everything it does with these tensors is ordered by the stream, like the reference's
            # tensors after its `.tolist()`.  Falls back to the blocking copy after 50 ms.
            self._khost_armed = False
            K = int(lib().vc2_wait_host_count(self._khost_addr, 0.05))
            if K >= 0:
                # the launch's FINAL status word arrives when its last selection ends (~10 us after K): taken if it is
                # there already, else this pass is settled before the plan's next one starts (_settle) -- a late
                # selection-guard hit is raised there, never lost
                fin = int(self._khost_final.value)
                if fin & _STATUS_FINAL:
                    status = fin & ~_STATUS_FINAL
                else:
                    status = int(self._khost.t[1])
                    self._late = (self.kout, self.cap, K)
                return self._result(self.take(), K, status)
        K, status = self.kout.tolist()                   # the single host sync of the path
        return self._result(self.take(), K, status)

    def _settle(self) -> None:
        """A pass that returned on the early status words of the host mirror: read its final status now (the device word,
        one blocking copy -- by now the launch is long over) and raise what it reports."""
        late, self._late = getattr(self, "_late", None), None
        if late is None:
            return
        kout, cap, K = late
        fin = int(self._khost_final.value) if self._khost is not None else 0
        status = (fin & ~_STATUS_FINAL) if fin & _STATUS_FINAL else int(kout[1].item())
        if status:
            try:
                _raise_status(status, cap, K)
            except RuntimeError as e:
                raise RuntimeError(f"the PREVIOUS pass of this plan (its result was already returned): {e}") from None

    def take(self):
        """The enqueued pass's output tensors, detached from the plan: the plan can take new outputs and the next clip
        (same stream: ordered) while this one is still running; `finish_many` reads the counts of a whole batch in one go."""
        return (self.rows, self.idx, self.ks, self.v, self.f, self.kout, self.cap, self.tail_rows)

    @staticmethod
    def _result(taken, K, status) -> CompressionResult:
        rows, idx, ks, v, f, _kout, cap, tail_rows = taken
        if status:
            _raise_status(int(status), cap, int(K))
        return CompressionResult(rows[:K + tail_rows] if rows is not None else None, idx[:K], ks, int(K), v, f)

    @staticmethod
    def finish_many(taken_list) -> List[CompressionResult]:
        """Results of several enqueued passes (CompressPlan.take) whose streams have been synchronised: ONE device-to-host
        copy for all their (K, status) words instead of a sync per clip."""
        if not taken_list:
            return []
        words = torch.stack([t[5] for t in taken_list]).tolist()
        return [CompressPlan._result(t, K, status) for t, (K, status) in zip(taken_list, words)]


# ---- plan cache of the one-shot API --------------------------------------------------------------------------
# `compress()` / `vidcom2_compression()` / the model hooks run one pass per call; building a CompressPlan (a ~30 MB
# workspace, status words) costs more than a quarter of the pass.  Plans are therefore kept per (shape, dtype, device,
# options, HIP stream, thread): work enqueued on ONE stream is ordered, so re-using a plan's workspace there is safe;
# another stream or thread gets its own plan.  Outputs are NOT cached (CompressPlan.new_outputs).
_PLAN_CACHE: "collections.OrderedDict" = collections.OrderedDict()
_PLAN_CACHE_MAX = int(os.environ.get("VC2_PLAN_CACHE", "8"))       # 0 disables the cache
_PLAN_CACHE_BYTES = int(os.environ.get("VC2_PLAN_CACHE_MB", "1024")) << 20   # workspaces + output sets kept alive by the cache
_SPARE_MAX_BYTES = int(os.environ.get("VC2_SPARE_MAX_MB", "256")) << 20      # no spare output set beyond this size
_PLAN_LOCK = threading.Lock()
_EARLY_COUNT = os.environ.get("VC2_EARLY_COUNT", "1") != "0"        # one-shot calls take K from a pinned host mirror (vc2_compress_ex2)
_PREALLOC = os.environ.get("VC2_PREALLOC", "1") != "0"             # next call's outputs allocated while this pass runs


def _cached_plan(F, N, D, dtype, device, base_scale, mapper, grid_h, want_scores, gather, tail_rows) -> "CompressPlan":
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    key = (F, N, D, dtype, device, float(base_scale), mapper, int(grid_h), bool(want_scores), bool(gather),
           int(tail_rows), torch.cuda.current_stream(device).cuda_stream, threading.get_ident(), _ffi.get_mode())
    # (the key holds the thread id, so a plan is only ever USED by one thread; the dictionary itself is shared: another
    #  thread's eviction between get and move_to_end raised KeyError before the lock)
    with _PLAN_LOCK:
        plan = _PLAN_CACHE.get(key) if _PLAN_CACHE_MAX > 0 else None
        if plan is not None:
            _PLAN_CACHE.move_to_end(key)
    if plan is None:
        plan = CompressPlan(F, N, D, dtype, device, base_scale, mapper, grid_h, want_scores, gather, tail_rows)
        if _PLAN_CACHE_MAX > 0:
            with _PLAN_LOCK:
                _PLAN_CACHE[key] = plan
                # bounded by entries AND by bytes (a long clip's workspace is hundreds of MB): oldest first
                # (a cached plan keeps its workspace AND its last output set, plus a spare one: all counted)
                evicted = []
                while len(_PLAN_CACHE) > _PLAN_CACHE_MAX or \
                        (len(_PLAN_CACHE) > 1 and sum(p.cached_bytes() for p in _PLAN_CACHE.values()) > _PLAN_CACHE_BYTES):
                    evicted.append(_PLAN_CACHE.popitem(last=False)[1])
            for old in evicted:               # (outside the lock: may read a device word)
                old.settle_quietly()          # a pass that returned on its early status words gets its last word here
    else:
        plan.new_outputs()
    return plan


def clear_plan_cache() -> None:
    with _PLAN_LOCK:
        plans = list(_PLAN_CACHE.values())
        _PLAN_CACHE.clear()
    for p in plans:
        p.settle_quietly()


atexit.register(clear_plan_cache)            # (the last call of the process: its final status word is read, not lost)


class _KHost:
    """Four pinned int64 words for vc2_compress_ex2's host mirror (K, early status, final status | 2^62, spare).  Pooled
    and never freed: the selection launch writes the final word ~10 us after the count, possibly after the plan that
    enqueued the pass is gone -- a block handed back to the allocator could be somebody else's memory by then."""
    __slots__ = ("t", "addr", "word", "final", "in_flight")

    def __init__(self):
        self.t = torch.zeros(4, dtype=torch.int64).pin_memory()
        self.addr = ctypes.c_void_p(self.t.data_ptr())
        self.word = ctypes.c_int64.from_address(self.t.data_ptr())
        self.final = ctypes.c_int64.from_address(self.t.data_ptr() + 16)
        self.in_flight = False


_KHOST_LOCK = threading.RLock()       # (re-entrant: a garbage collection inside the locked region may finalise a plan -> _khost_put)
_KHOST_FREE: list = []
_KHOST_QUARANTINE: list = []         # returned while their pass may still write the final word


def _khost_get() -> _KHost:
    with _KHOST_LOCK:
        still = []
        for b in _KHOST_QUARANTINE:
            (_KHOST_FREE if b.final.value & _STATUS_FINAL else still).append(b)
        _KHOST_QUARANTINE[:] = still
        b = _KHOST_FREE.pop() if _KHOST_FREE else None
    if b is None:
        b = _KHost()
    b.in_flight = False
    return b


def _khost_put(b: _KHost) -> None:
    with _KHOST_LOCK:
        (_KHOST_QUARANTINE if b.in_flight and not (b.final.value & _STATUS_FINAL) else _KHOST_FREE).append(b)


_STATUS_FINAL = 1 << 62          # K_host[2] of vc2_compress_ex2: "this is the launch's final status word"


def _raise_status(status: int, cap: int, K: int) -> None:
    """K_out[1] of a pass (include/vc2.h): every bit is a 'cannot happen' condition the kernels check for instead of
    returning a silently wrong kept set."""
    why = []
    if status & 1:
        why.append(f"kept-token capacity {cap} exceeded (K={K}): vc2_kept_capacity bound violated")
    if status & 2:
        why.append("a bounded wait between workgroups of one launch expired")
    if status & 4:
        why.append("a loop bound of the selection replay expired (vc2_selftest_counters has the details)")
    if status & ~7:
        why.append(f"unknown status bits {status:#x}")
    raise RuntimeError("vidcom2_amd: " + "; ".join(why) + " -- please report")


@_guarded
def compress(flattened_feat: torch.Tensor, tpf: int, base_scale: float = 0.25, mapper: str = "linear",
             grid_h: int = 0, img_feat: Optional[torch.Tensor] = None, want_scores: bool = False,
             gather: bool = True, tail: Optional[torch.Tensor] = None,
             stats_ws: Optional[torch.Tensor] = None) -> CompressionResult:
    """One whole pass: feature tensor resident in HBM -> kept rows + indices + budgets.
    tail ([m, D], e.g. LLaVA's newline embedding): written behind the kept rows by the gather launch itself;
    `rows` is then [K + m, D].  stats_ws: the workspace `fused.pool_stats` returned together with THIS tensor --
    the pass then skips its first sweep."""
    x = _prep(flattened_feat, "flattened_feat")
    if x.dim() != 2:
        raise RuntimeError(f"flattened_feat must be 2-D [frames*tokens, dim], got {tuple(x.shape)}")
    R, D = x.shape
    tpf = int(tpf)
    if tpf <= 0 or R % tpf != 0:
        # the reference fails in frames = x.view(-1, tpf, C) (vidcom2.py:47)
        raise RuntimeError(f"shape '[-1, {tpf}, {int(D * 0.5)}]' is invalid for input of size {R * int(D * 0.5)}")
    src = None
    if mapper == "grid_vid":
        if img_feat is None:
            raise ValueError("img_feat required for grid mapping")
        src = _prep(img_feat, "img_feat")
        if src.dtype != x.dtype or src.shape[-1] != D:
            raise RuntimeError("img_feat must have the dtype and feature dim of flattened_feat")
        need = (R // tpf) * int(grid_h) * (int(grid_h) + 1)
        if src.dim() != 2 or src.shape[0] < need:
            # the reference fails in img[_map_grid_vid(...)] (vidcom2.py:96) with an IndexError
            raise IndexError(f"index {need - 1} is out of bounds for dimension 0 with size {src.shape[0]}"
                             if src.dim() == 2 else "img_feat must be 2-D [rows, dim]")
    if tail is not None:
        tail = _prep(tail if tail.dim() == 2 else tail[None], "tail")
    if stats_ws is not None and (getattr(stats_ws, "_vc2_stats_for", None) != (R // tpf, tpf, D, x.dtype, x.data_ptr())
                                 or stats_ws.numel() < _ffi.workspace_bytes(R // tpf, tpf, D, x.dtype)):
        # another tensor (a copy was made), another (frames, tokens per frame) split -- e.g. a tpf that is not the
        # pooled grid -- or a workspace of unknown origin: its partials are not this call's, sweep 1 runs again
        stats_ws = None
    ntail = 0 if tail is None else tail.shape[0]
    if stats_ws is not None:          # (the statistics live in the caller's workspace: a plan around it, not cached)
        plan = CompressPlan(R // tpf, tpf, D, x.dtype, x.device, base_scale, mapper, grid_h, want_scores, gather,
                            tail_rows=ntail, ws=stats_ws)
    else:
        plan = _cached_plan(R // tpf, tpf, D, x.dtype, x.device, base_scale, mapper, grid_h, want_scores, gather, ntail)
    plan.enqueue(x, src, tail, have_stats=stats_ws is not None, mirror=_EARLY_COUNT)
    if stats_ws is None and _PREALLOC and _PLAN_CACHE_MAX > 0 and plan.output_bytes() <= _SPARE_MAX_BYTES:
        plan.prepare_spare()              # (the GPU is busy for the next ~190 us: the next call's outputs cost nothing here)
    return plan.finish()


_LANES: dict = {}
_LANES_LOCK = threading.Lock()


def _lane_streams(dev: torch.device, n: int):
    """The n extra streams compress_batch runs clips on -- created once per (device, thread) and kept: the plan cache is
    keyed by stream, so fresh streams per call meant fresh plans (and workspaces) per call: 0.5-1.2 ms of set-up per batch."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), threading.get_ident())
    with _LANES_LOCK:
        if key not in _LANES:                 # a new thread: drop the lanes of threads that have ended
            alive = {t.ident for t in threading.enumerate()}
            for k in [k for k in _LANES if k[1] not in alive]:
                del _LANES[k]
        lst = _LANES.setdefault(key, [])
        while len(lst) < n:
            lst.append(torch.cuda.Stream(dev))
        return lst[:n]


BATCH_LANES = 3          # compress_batch's default: clips in flight (16 cfg5 clips: 254 / 216 / 203 / 197 us per clip with 1 / 2 / 3 / 4)


def compress_batch(clips, tpf: int, base_scale: float = 0.25, in_flight: int = BATCH_LANES, gather: bool = True,
                   own_storage: bool = False):
    """Compress several clips (a list of [F_i * tpf, D] tensors) with up to `in_flight` of them running
    concurrently, one HIP stream each.  A single pass leaves the GPU idle during its single-workgroup selection
    replays; further clips in flight fill those gaps (DESIGN.md "clips in flight"; every lane keeps a plan = a workspace).
    Returns one CompressionResult per clip, in order.  The reference has no batched form: its harness loops
    over clips (lmms-eval, batch size 1 per rank).
    Storage: for clips of ONE shape every result is a VIEW into batch-wide buffers (one allocation per output kind), so
    keeping any clip's rows / indices alive keeps the whole batch's buffers alive (n * capacity * D elements);
    own_storage=True hands every clip its own tensors instead (one extra device copy of the kept rows)."""
    clips = [_prep(c, "clip") for c in clips]
    if not clips:
        return []
    dev = clips[0].device
    for i, x in enumerate(clips):
        if x.dim() != 2 or tpf <= 0 or x.shape[0] % int(tpf) != 0:
            raise RuntimeError(f"clip {i}: shape {tuple(x.shape)} is not [frames * {tpf}, dim]")
    cur = torch.cuda.current_stream(dev)
    lanes = [cur] + _lane_streams(dev, max(1, min(int(in_flight), len(clips))) - 1)
    same = all(c.shape == clips[0].shape and c.dtype == clips[0].dtype and c.device == dev for c in clips)
    if not same:
        return _compress_batch_mixed(clips, int(tpf), base_scale, lanes, gather)
    # Clips of one shape (the batched-eval case, BASELINE config 5): ONE plan per lane (workspace; work on a stream is
    # ordered), ONE allocation per output kind for the whole batch -- each clip's pass writes its slice --, one
    # device-to-host copy for all the counts at the end.  Round 4 looked a plan up, allocated six tensors and recorded
    # them on the caller's stream PER CLIP: ~100 us of host work per clip, more than the 75 us a lane's share of the GPU
    # time leaves -- 16 clips took 5.8 ms, slower than one at a time.
    n, x0 = len(clips), clips[0]
    F_, D_ = x0.shape[0] // int(tpf), x0.shape[1]
    plans = []
    for st in lanes:
        with torch.cuda.stream(st):
            plans.append(_cached_plan(F_, int(tpf), D_, x0.dtype, dev, base_scale, "linear", 0, False, gather, 0))
    p0 = plans[0]
    idx_all = torch.empty((n, p0.cap), dtype=torch.int64, device=dev)
    ks_all = torch.empty((n, F_), dtype=torch.int64, device=dev)
    rows_all = torch.empty((n, p0.cap, D_), dtype=x0.dtype, device=dev) if gather else None
    kout_all = torch.empty((n, 2), dtype=torch.int64, device=dev)
    for st in lanes[1:]:
        st.wait_stream(cur)               # inputs and these buffers belong to the caller's stream
    # (no record_stream on the batch's buffers: every lane is joined into the caller's stream below, before this function
    #  returns -- whatever the caller does with them, freeing included, is ordered behind the lanes' work)
    for i, x in enumerate(clips):
        plan = plans[i % len(lanes)]
        plan.idx, plan.ks, plan.rows, plan.v, plan.f, plan.kout = idx_all[i], ks_all[i], (rows_all[i] if gather else None), None, None, kout_all[i]
        plan.enqueue(x, stream=lanes[i % len(lanes)])
    for st in lanes[1:]:
        cur.wait_stream(st)               # results are safe to use on the current stream
    for plan in plans:
        plan.new_outputs()                # (the cached plans must not keep pointing into the batch's buffers)
    words = kout_all.tolist()             # the single host sync of the batch
    out = []
    for i, (K, status) in enumerate(words):
        if status:
            _raise_status(int(status), p0.cap, int(K))
        rows_i, idx_i, ks_i = (rows_all[i, :K] if gather else None), idx_all[i, :K], ks_all[i]
        if own_storage:
            rows_i, idx_i, ks_i = (rows_i.clone() if gather else None), idx_i.clone(), ks_i.clone()
        out.append(CompressionResult(rows_i, idx_i, ks_i, int(K), None, None))
    return out


def _compress_batch_mixed(clips, tpf: int, base_scale: float, lanes, gather: bool):
    """compress_batch for clips of different shapes: a plan lookup and fresh outputs per clip."""
    dev = clips[0].device
    cur = lanes[0]
    for st in lanes[1:]:
        st.wait_stream(cur)                               # inputs were produced on the current stream
    taken = []
    for i, x in enumerate(clips):
        st = lanes[i % len(lanes)]
        with torch.cuda.stream(st):       # the plan's buffers are allocated (and owned) on the lane's stream
            plan = _cached_plan(x.shape[0] // int(tpf), int(tpf), x.shape[1], x.dtype, dev, base_scale, "linear", 0,
                                False, gather, 0)
            plan.enqueue(x)
            taken.append(plan.take())
            if st is not cur:             # allocated on the lane's stream, consumed on the caller's
                for t in taken[-1][:6]:
                    if t is not None:
                        t.record_stream(cur)
    for st in lanes[1:]:
        cur.wait_stream(st)                               # results are safe to use on the current stream
    return CompressPlan.finish_many(taken)


@_guarded
def vidcom2_compression(flattened_feat: torch.Tensor, model: str = "llava_ov", base_scale: float = 0.25,
                        frame_token_len: Optional[int] = None,
                        img_feat: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Reference: vidcom2.py:15-36.  Returns the gathered kept rows ([K, D], input dtype)."""
    if model not in MODEL_SPECS:
        raise ValueError(f"Unknown model: {model}")
    spec = MODEL_SPECS[model]
    tpf = frame_token_len if model in _DYNAMIC_TPF else spec["tpf"]
    if tpf is None:
        raise ValueError(f"frame_token_len required for {model}")
    tpf = _as_int(tpf)
    if spec["mapper"] == "grid_vid":
        if img_feat is None:
            # the reference scores first and raises in map_features (vidcom2.py:94); same exception
            raise ValueError("img_feat required for grid mapping")
        return compress(flattened_feat, tpf, base_scale, "grid_vid", spec["grid"], img_feat).rows
    return compress(flattened_feat, tpf, base_scale, "linear").rows


# ---------------------------------------------------------------------------------------------
# stage functions (same decomposition as the reference)
# ---------------------------------------------------------------------------------------------

@_guarded
def _channel_variance(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x.var(dim=0, unbiased=False) (vidcom2.py:40): returns (var in T, fp32-widened copy)."""
    R, D = x.shape
    ws = _ffi.workspace(1, R, D, x.dtype, x.device)
    var_T = torch.empty(D, dtype=x.dtype, device=x.device)
    var_f = torch.empty(D, dtype=torch.float32, device=x.device)
    check(lib().vc2_chan_var(ptr(x), R, D, DTYPE_CODE[x.dtype], ptr(ws), ws.numel(), ptr(var_T), ptr(var_f),
                             stream_ptr(x.device)), "vc2_chan_var")
    return var_T, var_f


@_guarded
def low_var_channel_order(x: torch.Tensor, ratio: float = 0.5) -> torch.Tensor:
    """Indices torch.topk(var, int(D*ratio), largest=False) returns on the CPU reference, in ITS order
    (ascending variance, libstdc++ nth_element + sort tie order), replayed on the device
    (k_chan_select + k_chan_order: workgroup-parallel introselect + introsort)."""
    x = _prep(x, "x")
    D = x.shape[-1]
    k = int(D * ratio)
    _, var_f = _channel_variance(x)
    order = torch.empty(max(k, 1), dtype=torch.int32, device=x.device)
    perm = torch.empty(max(k, 1), dtype=torch.int32, device=x.device)
    if k > 0:
        check(lib().vc2_chan_select(ptr(var_f), D, k, None, None, ptr(perm), ptr(order), None, None,
                                    stream_ptr(x.device)), "vc2_chan_select")
    return order[:k].to(torch.int64)


@_guarded
def select_low_var_channels(x: torch.Tensor, ratio: float = 0.5) -> torch.Tensor:
    """Reference: vidcom2.py:38-43.  Returns x[:, idx] -- a copy, columns in topk order."""
    x = _prep(x, "x")
    if x.dim() != 2:
        raise RuntimeError("select_low_var_channels expects a 2-D [tokens, dim] tensor")
    idx = low_var_channel_order(x, ratio)
    R, D = x.shape
    C = idx.numel()
    out = torch.empty((R, C), dtype=x.dtype, device=x.device)
    if R and C:
        check(lib().vc2_gather_cols(ptr(x), R, D, DTYPE_CODE[x.dtype], ptr(idx), C, ptr(out), stream_ptr(x.device)),
              "vc2_gather_cols")
    return out


@_guarded
def compute_gaussian_scores(x: torch.Tensor, tpf: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference: vidcom2.py:45-57.  x = channel-selected features [F*tpf, C]; returns (v, f) [F, tpf]."""
    x = _prep(x, "x")
    tpf = _as_int(tpf)
    R, C = x.shape
    if tpf <= 0 or R % tpf != 0:
        raise RuntimeError(f"shape '[-1, {tpf}, {C}]' is invalid for input of size {R * C}")
    F = R // tpf
    ws = _ffi.workspace(F, tpf, C, x.dtype, x.device)
    v = torch.empty((F, tpf), dtype=x.dtype, device=x.device)
    f = torch.empty((F, tpf), dtype=x.dtype, device=x.device)
    check(lib().vc2_scores(ptr(x), F, tpf, C, DTYPE_CODE[x.dtype], None, C, None, ptr(ws), ws.numel(), ptr(v),
                           ptr(f), None, None, stream_ptr(x.device)), "vc2_scores")
    return v, f


def _multi_scale_gaussian(x: torch.Tensor, center: torch.Tensor, alphas: List[float]) -> torch.Tensor:
    """Reference: vidcom2.py:59-62.  x T[F, N, C], center T[1, 1, C] or T[F, 1, C] (the two shapes
    compute_gaussian_scores uses, vidcom2.py:51-52) -> T[F, N].  The fused pass never calls this (it
    does not materialise x); it exists for callers of the helper itself."""
    alphas = [float(a) for a in alphas]
    if not alphas:
        return 0                     # Python's sum() over no terms, exactly like the reference
    # (x - center) broadcasts like torch: any x [..., C] and any center broadcastable to it
    out_shape = torch.broadcast_shapes(tuple(x.shape), tuple(center.shape))
    if len(out_shape) < 1:
        raise RuntimeError("_multi_scale_gaussian: x must have a channel dimension")
    C = out_shape[-1]
    xb = x.expand(out_shape)
    cb = center.to(x.dtype).expand(out_shape)
    lead = out_shape[:-1]
    R = 1
    for d in lead:
        R *= int(d)
    # centre layouts the kernel knows: one row, one row per "frame" (constant along the last leading dim), or one
    # row per token (anything else)
    if all(int(cb.stride(i)) == 0 or out_shape[i] == 1 for i in range(len(lead))):
        F, N, cc = 1, R, cb[tuple(0 for _ in lead)].reshape(1, C)
    elif len(lead) >= 2 and (int(cb.stride(len(lead) - 1)) == 0 or out_shape[len(lead) - 1] == 1):
        N = int(out_shape[len(lead) - 1])
        F = R // N
        cc = cb.select(len(lead) - 1, 0).reshape(F, C)
    else:
        F, N, cc = R, 1, cb.reshape(R, C)
    xx = _prep(xb.reshape(R, C), "x")
    cc = _prep(cc, "center")
    out = torch.empty(lead, dtype=x.dtype, device=x.device)
    if R == 0:
        return out
    arr = (ctypes.c_double * len(alphas))(*alphas)
    with on_device(x.device):
        check(lib().vc2_multi_scale_gaussian(ptr(xx), F, N, C, DTYPE_CODE[x.dtype], ptr(cc), cc.shape[0], arr,
                                             len(alphas), ptr(out), stream_ptr(x.device)), "_multi_scale_gaussian")
    return out


@_guarded
def compute_scales(scores: torch.Tensor, base: float, temp: float = 0.01) -> torch.Tensor:
    """Reference: vidcom2.py:64-68.  scores T[F] -> scales T[F]."""
    s = _prep(scores, "scores")
    if s.dim() != 1:
        raise RuntimeError("compute_scales expects a 1-D tensor of per-frame scores")
    F = s.numel()
    out = torch.empty_like(s)
    if F == 0:
        return out
    ws = torch.empty(3 * (F * 4 + 256), dtype=torch.uint8, device=s.device)
    check(lib().vc2_compute_scales(ptr(s), F, float(base), float(temp), DTYPE_CODE[s.dtype], ptr(ws), ws.numel(),
                                   ptr(out), stream_ptr(s.device)), "vc2_compute_scales")
    return out


def _select(scores: torch.Tensor, scales: torch.Tensor, tpf: int, map_mode: int, grid_h: int = 0):
    sc = _prep(scores, "scores")
    sl = _prep(scales, "scales")
    if sc.dim() != 2 or sl.dim() != 1 or sl.numel() != sc.shape[0]:
        raise RuntimeError("scores must be [F, N] and scales [F]")
    if sl.dtype != sc.dtype:
        sl = sl.to(sc.dtype)
    F, N = sc.shape
    extra = grid_h if map_mode == MAP_GRID_VID else 0
    cap = F * (N + extra)
    ws = torch.empty(F * N * 4 + F * 4 + 1024, dtype=torch.uint8, device=sc.device)
    ks = torch.empty(F, dtype=torch.int64, device=sc.device)
    offs = torch.empty(F + 1, dtype=torch.int64, device=sc.device)
    idx = torch.empty(cap, dtype=torch.int64, device=sc.device)
    kout = torch.empty(2, dtype=torch.int64, device=sc.device)
    with on_device(sc.device):
        check(lib().vc2_select(ptr(sc), ptr(sl), F, N, int(tpf), DTYPE_CODE[sc.dtype], map_mode, grid_h, ptr(ws),
                               ws.numel(), ptr(ks), ptr(offs), ptr(idx), cap, ptr(kout), stream_ptr(sc.device)),
              "vc2_select")
    return idx, ks, offs, kout


def select_outlier_indices(scores: torch.Tensor, scales: torch.Tensor, tpf: int) -> List[torch.Tensor]:
    """Reference: vidcom2.py:70-78.  Returns F ascending int64 index tensors (one host sync, as the
    reference's ``.tolist()``)."""
    idx, ks, _, kout = _select(scores, scales, _as_int(tpf), MAP_LOCAL)
    host = torch.cat((ks, kout)).tolist()     # budgets + (count, status): still ONE host sync
    ks_host, (K, status) = host[:-2], host[-2:]
    if status:
        _raise_status(int(status), int(idx.numel()), int(K))
    N = scores.shape[1]
    if any(k > N for k in ks_host):          # ks = round(scales * tpf) with tpf > N: torch.topk's own error (vidcom2.py:76)
        raise RuntimeError("selected index k out of range")
    return list(torch.split(idx[: sum(ks_host)], ks_host))


def _cat_indices(indices: List[torch.Tensor]):
    if len(indices) == 0:
        raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")
    dev = indices[0].device
    require_device(indices[0], "indices[0]")
    ks_host = [int(i.numel()) for i in indices]
    loc = torch.cat([i.to(torch.int64) for i in indices]).contiguous()
    ks = torch.tensor(ks_host, dtype=torch.int64)
    offs = torch.zeros(len(indices) + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(ks, 0)
    return loc, ks.to(dev), offs.to(dev), ks_host


@_guarded
def _map_linear_offset(indices: List[torch.Tensor], tpf: int) -> torch.Tensor:
    """Reference: vidcom2.py:99-103."""
    loc, ks, offs, _ = _cat_indices(indices)
    out = torch.empty_like(loc)
    if loc.numel():
        check(lib().vc2_map_indices(ptr(loc), ptr(ks), ptr(offs), len(indices), MAP_LINEAR, _as_int(tpf), ptr(out),
                                    stream_ptr(loc.device)), "vc2_map_indices")
    return out


@_guarded
def _map_grid_vid(indices: List[torch.Tensor], h: int) -> torch.Tensor:
    """Reference: vidcom2.py:105-115 (per frame: kept tokens on the h x (h+1) grid, then its h newlines)."""
    loc, ks, offs, ks_host = _cat_indices(indices)
    h = _as_int(h)
    out = torch.empty(loc.numel() + len(indices) * h, dtype=torch.int64, device=loc.device)
    check(lib().vc2_map_indices(ptr(loc), ptr(ks), ptr(offs), len(indices), MAP_GRID_VID, h, ptr(out),
                                stream_ptr(loc.device)), "vc2_map_indices")
    return out


@_guarded
def _gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    src = _prep(src, "features")
    K = idx.numel()
    out = torch.empty((K,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if K == 0:
        return out
    D = src[0].numel()
    kdev = torch.tensor([K, 0], dtype=torch.int64, device=src.device)
    check(lib().vc2_gather_rows(ptr(src), src.shape[0], D, DTYPE_CODE[src.dtype], ptr(idx), ptr(kdev), K, ptr(out),
                                stream_ptr(src.device)), "vc2_gather_rows")
    return out


@_guarded
def map_features(indices: List[torch.Tensor], flat: torch.Tensor, img: Optional[torch.Tensor],
                 spec: Dict[str, Any]) -> torch.Tensor:
    """Reference: vidcom2.py:80-97."""
    if spec["mapper"] == "linear":
        stride = flat.shape[0] // len(indices)                     # vidcom2.py:89
        return _gather_rows(flat, _map_linear_offset(indices, stride))
    elif spec["mapper"] == "grid_vid":
        if img is None:
            raise ValueError("img_feat required for grid mapping")
        return _gather_rows(img, _map_grid_vid(indices, spec["grid"]))
    return flat
