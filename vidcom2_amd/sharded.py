"""Frame-sharded VidCom2 compression across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank p holds frames
[p*F_local, (p+1)*F_local) of ONE video.  The pass shards by frame with three tiny exchange steps --
every message is <= a few hundred KB, i.e. latency-bound, so each is a single all-gather followed by
a fixed-order local reduction (NOT an all-reduce: the fixed order makes every rank compute the same
bits; the partials are CANONICAL -- cut at fixed frame boundaries and folded in one fixed order -- so the bits are also
the same for every world size, and the same as the unsharded pass's, when each rank holds a multiple of 16 frames):

  after sweep 1   per 8-frame block statistics (mean, M2)  fp64 [F_l/8, 2, D] -> identical channel mask
  after sweep 2   per 16-frame sums of normalised tokens   fp64 [F_l/16, C]   -> identical video centre
  after sweep 3   per-frame uniqueness scores  -mean(v)    fp32 [F_local] -> global softmax budgets
                  (the RCCL all-gather BASELINE.json's north_star names)

Kept rows stay sharded by frame (each rank returns the kept rows / indices of its own frames); the
reference has no multi-GPU form of this path -- its only parallelism is document-level DP in the
harness (lmms-eval/lmms_eval/evaluator.py:488-491), which is the "replicas only" case.

The arithmetic lives behind a small stage interface so that the collective logic can be exercised on
CPU with gloo (tests inject an oracle-backed stage object); the default stages are the HIP kernels.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from . import _ffi
from ._ffi import DTYPE_CODE, check, lib, on_device, ptr, stream_ptr


@dataclass
class ShardResult:
    rows: Optional[torch.Tensor]     # [K_local, D] kept rows of this rank's frames
    local_idx: torch.Tensor          # int64 [K_local]  f_local*N + n
    global_idx: torch.Tensor         # int64 [K_local]  index into the whole video's [F_total*N] tokens
    ks: torch.Tensor                 # int64 [F_local]
    K: int


class HipStages:
    """The five shard-local stages on the HIP kernels (include/vc2.h 'frame-sharded building blocks')."""

    def __init__(self, F: int, N: int, D: int, dtype, device, base_scale: float, gather: bool = True,
                 vc_cap: int = 64):
        self.F, self.N, self.D, self.dtype, self.device, self.base = F, N, D, dtype, torch.device(device), base_scale
        self.code = DTYPE_CODE[dtype]
        L = lib()
        self.ws = _ffi.workspace(F, N, D, dtype, self.device)
        # canonical partials: per stat block of `bf` frames (mean, M2); per group of 16 frames the
        # sums of the normalised tokens.  bf = 8 whenever F is a multiple of 8 (then the gathered blocks are exactly
        # the unsharded pass's blocks and every world size reduces to the same bits)
        self.bf = next(b for b in (8, 4, 2, 1) if F % b == 0)
        self.stats = torch.empty((F // self.bf, 2, D), dtype=torch.float64, device=self.device)
        self.C = int(D * 0.5)                                   # int(x.shape[-1] * ratio), vidcom2.py:41
        # exchange 2: per group of 16 frames the sums of x^ and, behind them, the groups' bounds of sum |x^|
        self.csum = torch.zeros((2 * ((F + 15) // 16), self.C), dtype=torch.float64, device=self.device)
        self.F_total = F                                        # set by ShardedCompressor (world * F)
        self.var_f32 = torch.empty(D, dtype=torch.float32, device=self.device)
        self.mask = torch.empty(D, dtype=torch.uint8, device=self.device)
        self.cols = torch.empty(D, dtype=torch.int32, device=self.device)
        self.perm = torch.empty(D, dtype=torch.int32, device=self.device)     # channels as nth_element left them
        self.spos = torch.empty(D, dtype=torch.int32, device=self.device)     # their positions in that order
        self.total = torch.empty(F * N, dtype=torch.float32, device=self.device)
        self.s = torch.empty(F, dtype=torch.float32, device=self.device)
        # a rank can hold the globally dominant frame: sum of its scales <= base * (F_local + 1)
        self.cap = min(F * N, int(L.vc2_kept_capacity(F + 1, N, base_scale)))
        self.idx = torch.empty(self.cap, dtype=torch.int64, device=self.device)
        self.ks = torch.empty(F, dtype=torch.int64, device=self.device)
        self.kout = torch.zeros(4, dtype=torch.int64, device=self.device)     # K, capacity overflow, fragile centre columns
        # exchange 2b (video-centre replay): level-0 block sums of this rank's rows for up to vc_cap flagged columns
        self.vc_cap = int(vc_cap)                             # flagged columns whose blocks one exchange carries
        self.vc_replay = dtype != torch.float32
        # per flagged column: 64 + 64 raw edge values and the sums of this rank's complete blocks (include/vc2.h)
        self.blocks = torch.zeros((self.vc_cap, 129 + F * N // 16), dtype=torch.float32, device=self.device)
        self.rows = torch.empty((self.cap, D), dtype=dtype, device=self.device) if gather else None

        # device pointers of the fixed buffers, resolved once: the per-pass host work is five C calls plus
        # three collectives, and that host time bounds the pass when it exceeds the ~0.3 ms of GPU work
        self._L = L
        self._p = {n: ptr(getattr(self, n)) for n in ("ws", "stats", "csum", "var_f32", "mask", "cols", "perm",
                                                     "spos", "total", "s", "idx", "ks", "kout", "rows", "blocks")}
        self._ws_n = self.ws.numel()

    def _st(self):
        return stream_ptr(self.device)

    def chan_stats(self, x):
        p = self._p
        check(self._L.vc2_chan_stats(ptr(x), self.F, self.N, self.D, self.code, self.F_total, self.bf, p["ws"],
                                     self._ws_n, p["stats"], self._st()), "vc2_chan_stats")
        return self.stats

    def select_channels(self, stats_all, R_total):
        p, st = self._p, self._st()
        blocks = stats_all.reshape(-1, 2, self.D)               # [world * nb, 2, D], rank order = frame order
        check(self._L.vc2_chan_var_from_stats(ptr(blocks), blocks.shape[0], self.bf * self.N, R_total, self.D, self.code,
                                              None, p["var_f32"], st), "vc2_chan_var_from_stats")
        # the channel SET now; torch.topk's channel ORDER (needed only by the fix-up kernels of phase 1) is
        # replayed from `perm` by a rider workgroup of sweep 2 inside vc2_scores_phase1
        check(self._L.vc2_chan_select(p["var_f32"], self.D, self.C, p["mask"], p["cols"], p["perm"], None, None, None,
                                      st), "vc2_chan_select")

    def phase1(self, x):
        p = self._p
        check(self._L.vc2_scores_phase1(ptr(x), self.F, self.N, self.D, self.code, p["cols"], self.C, p["spos"],
                                        p["perm"], p["var_f32"], self.F_total, p["ws"], self._ws_n, p["csum"],
                                        self._st()),
              "vc2_scores_phase1")
        return self.csum

    def vc_blocks(self, x, csum_all, R_total, f0=0):
        """Exchange 2b: this rank's level-0 block sums (and the raw values of the blocks it shares with its neighbours)
        of the boundary-near video-centre columns, or None when the replay does not apply (fp32, exact mode, more tokens
        than the replay models) -- a decision every rank takes identically.  f0 = the video index of this rank's first
        frame.  A block may meet any number of ranks (since round 4: ranks with fewer rows than a block included)."""
        from .vidcom2 import cascade_modelled
        Rl = self.F * self.N
        if not self.vc_replay or _ffi.get_mode() != "torch" or not cascade_modelled(R_total) or R_total % Rl != 0:
            return None
        p = self._p
        parts = csum_all.reshape(-1, csum_all.shape[-1])
        check(self._L.vc2_video_centre_blocks(ptr(x), self.F, self.N, self.D, self.code, p["cols"], self.C, p["spos"],
                                              ptr(parts), parts.shape[0], parts.shape[1], self.csum.shape[0], R_total,
                                              int(f0) * self.N, p["ws"], self._ws_n, p["blocks"], self.vc_cap,
                                              self._st()),
              "vc2_video_centre_blocks")
        return self.blocks

    def flagged_columns(self) -> int:
        """How many video-centre columns the flag kernel of vc_blocks marked (SYNCHRONISES: one device-to-host read; every
        rank reads the same number -- the flags come from the all-gathered sums).  Only the modes that can flag more
        columns than one exchange carries ask (proven margins, debug)."""
        out = ctypes.c_int32(0)
        check(self._L.vc2_video_centre_flagged(self.F, self.N, self.D, self.code, self.F_total * self.N, self._p["ws"],
                                               self._ws_n, ctypes.byref(out), self._st()), "vc2_video_centre_flagged")
        return int(out.value)

    def needs_rounds(self) -> bool:
        """Modes whose video-centre margins can flag more than vc_cap columns: 3 (proven) and 2 (debug: all of them)."""
        return self.vc_replay and int(self._L.vc2_get_mode()) in (2, 3)

    def vc_blocks_round(self, x, R_total, f0, col_offset):
        p = self._p
        check(self._L.vc2_video_centre_blocks_round(ptr(x), self.F, self.N, self.D, self.code, p["cols"], self.C, p["spos"],
                                                    R_total, int(f0) * self.N, p["ws"], self._ws_n, p["blocks"], self.vc_cap,
                                                    int(col_offset), self._st()), "vc2_video_centre_blocks_round")
        return self.blocks

    def vc_finish_round(self, R_total, blocks_all, col_offset):
        p = self._p
        check(self._L.vc2_video_centre_finish_round(self.F, self.N, self.D, self.code, self.C, p["spos"], R_total, p["ws"],
                                                    self._ws_n, ptr(blocks_all), int(blocks_all.shape[0]), self.vc_cap,
                                                    int(col_offset), self._st()), "vc2_video_centre_finish_round")

    def phase2(self, x, csum_all, R_total, blocks_all=None, vc_final=False):
        """vc_final: the video centre in the workspace is final (vc_blocks + vc_finish_round rounds): sweep 3 only."""
        p = self._p
        parts = csum_all.reshape(-1, csum_all.shape[-1])       # [world * groups, C], rank order = frame order
        world = -1 if vc_final else (0 if blocks_all is None else int(blocks_all.shape[0]))
        if vc_final:
            blocks_all = None
        check(self._L.vc2_scores_phase2_blocks(ptr(x), self.F, self.N, self.D, self.code, p["cols"], self.C, p["spos"],
                                               ptr(parts), parts.shape[0], parts.shape[1], self.csum.shape[0], R_total,
                                               p["ws"], self._ws_n, None, None, p["total"], p["s"],
                                               ptr(blocks_all) if blocks_all is not None else None, world, self.vc_cap,
                                               self._st()), "vc2_scores_phase2")
        return self.s

    def select(self, x, s_all, f0):
        p = self._p
        check(self._L.vc2_select_sharded(p["total"], ptr(s_all), s_all.numel(), f0, self.F, self.N, self.D,
                                         float(self.base), self.code, p["ws"], self._ws_n, p["ks"], p["idx"], self.cap,
                                         p["kout"], ptr(x if self.rows is not None else None), p["rows"], self._st()),
              "vc2_select_sharded")

    def result(self, f0):
        K, overflow, vc_fragile, _ = self.kout.tolist()     # the path's single host sync
        if overflow:
            from .vidcom2 import _raise_status
            _raise_status(int(overflow), self.cap, int(K))
        self.vc_fragile = int(vc_fragile)
        if vc_fragile and _ffi.get_mode() == "torch":
            import warnings
            warnings.warn(f"vidcom2_amd (frame-sharded pass): {vc_fragile} video-centre value(s) lie within the replay margin "
                          "of a rounding boundary and could not be replayed across ranks "
                          f"(more than {self.vc_cap} such columns in a mode that does not go in rounds, or a channel count that is not a multiple of 32); they keep "
                          "the exactly rounded mean, the reference's fp32 summation order could round the other way.",
                          RuntimeWarning, stacklevel=2)
        li = self.idx[:K]
        return ShardResult(self.rows[:K] if self.rows is not None else None, li, li + f0 * self.N, self.ks, int(K))


def _all_gather(t: torch.Tensor, group, world: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[world, *t.shape] stacked in rank order (one all-gather; RCCL on GPU tensors, gloo on CPU).  `out` is a
    preallocated [world, *t.shape] buffer (the hot path passes one; tests may not)."""
    if world == 1:
        return t.unsqueeze(0)
    if out is None:
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=group)
    return out


class ShardedCompressor:
    """Frame-sharded pass; every rank calls enqueue(x_local) + finish() collectively."""

    def __init__(self, F_local: int, N: int, D: int, dtype, device, base_scale: float = 0.25, group=None,
                 stages=None, gather: bool = True, always_collective: bool = False):
        """always_collective: issue the all-gathers even at world size 1 (a process group must be initialised) -- the
        only way to put the RCCL exchanges themselves under test on a single-GPU box."""
        self.group = group
        self.always_collective = bool(always_collective)
        self._xev = None                                  # per-exchange hipEvent pairs (profile_exchanges)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.F, self.N, self.D = int(F_local), int(N), int(D)
        if self.world > 1:
            # every rank must bring the same number of frames (the canonical partials, the exchanges' message sizes and
            # the replay blocks are cut on that assumption): say so instead of hanging in a mismatched collective
            fl = [None] * self.world
            dist.all_gather_object(fl, (self.F, self.N, self.D), group=group)
            if any(t != fl[0] for t in fl):
                raise ValueError(f"ShardedCompressor: every rank must hold the same (frames, tokens per frame, width); got {fl}")
        self.F_total = self.F * self.world
        self.f0 = self.rank * self.F
        self.stages = stages if stages is not None else HipStages(self.F, self.N, self.D, dtype, device, base_scale,
                                                                  gather)
        if isinstance(self.stages, HipStages):
            self.stages.F_total = self.F_total

        self._gbuf = {}

    def _gather(self, key: str, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self.always_collective:
            return t.unsqueeze(0)
        buf = self._gbuf.get(key)
        if buf is None or buf.shape[1:] != t.shape or buf.dtype != t.dtype or buf.device != t.device:
            buf = self._gbuf[key] = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if self._xev is None or not t.is_cuda:
            return self._collect(t, buf)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = self._collect(t, buf)
        b.record()
        self._xev.setdefault(key, []).append((a, b))
        return out

    def _collect(self, t: torch.Tensor, buf: torch.Tensor) -> torch.Tensor:
        if self.world == 1:                                # (always_collective: the collective itself, one rank)
            dist.all_gather_into_tensor(buf.view(-1), t.contiguous().view(-1), group=self.group)
            return buf
        return _all_gather(t, self.group, self.world, buf)

    def profile_exchanges(self, on: bool = True) -> None:
        """hipEvents around every exchange of the passes that follow (stats / csum / blocks / s)."""
        self._xev = {} if on else None

    def exchange_times_us(self) -> dict:
        """Mean microseconds per exchange since profile_exchanges(True) (synchronises); the message sizes with them."""
        torch.cuda.synchronize()
        out = {}
        for key, evs in (self._xev or {}).items():
            buf = self._gbuf.get(key)
            out[key] = {"us": round(sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e3, 1),
                        "bytes_per_rank": int(buf[0].numel() * buf.element_size()) if buf is not None else None}
        return out

    def enqueue(self, x_local: torch.Tensor) -> None:
        st = self.stages
        if isinstance(st, HipStages):
            if x_local.dim() != 2 or tuple(x_local.shape) != (self.F * self.N, self.D) or x_local.dtype != st.dtype \
                    or x_local.device != st.device:
                raise RuntimeError(f"x_local must be a {st.dtype} [{self.F * self.N}, {self.D}] tensor on {st.device}, "
                                   f"got {x_local.dtype} {tuple(x_local.shape)} on {x_local.device}")
            if not x_local.is_contiguous():
                x_local = x_local.contiguous()
            with on_device(st.device):
                return self._enqueue(x_local)
        return self._enqueue(x_local)

    def _enqueue(self, x_local: torch.Tensor) -> None:
        st = self.stages
        R_total = self.F_total * self.N
        stats_all = self._gather("stats", st.chan_stats(x_local))               # exchange 1: [W, 2, D] fp64
        st.select_channels(stats_all, R_total)
        csum_all = self._gather("csum", st.phase1(x_local))                     # exchange 2: [W, D] fp64
        blocks = st.vc_blocks(x_local, csum_all, R_total, self.f0) if hasattr(st, "vc_blocks") else None
        if blocks is not None and getattr(st, "needs_rounds", lambda: False)():
            # proven-margin / debug modes: as many rounds of exchange 2b as the flagged columns need (one host sync for
            # their count -- identical on every rank, so every rank issues the same collectives)
            n = st.flagged_columns()
            for j0 in range(0, max(n, 1), st.vc_cap):
                if j0:
                    blocks = st.vc_blocks_round(x_local, R_total, self.f0, j0)
                st.vc_finish_round(R_total, self._gather("blocks", blocks), j0)
            s_loc = st.phase2(x_local, csum_all, R_total, vc_final=True)
        elif blocks is not None:                                                 # exchange 2b: [W, vc_cap, 129 + R_local/16] fp32
            s_loc = st.phase2(x_local, csum_all, R_total, self._gather("blocks", blocks))
        else:
            s_loc = st.phase2(x_local, csum_all, R_total)
        s_all = self._gather("s", s_loc)                                         # exchange 3: [W, F_local]
        st.select(x_local, s_all.reshape(-1), self.f0)

    def finish(self) -> ShardResult:
        return self.stages.result(self.f0)

    def __call__(self, x_local: torch.Tensor) -> ShardResult:
        self.enqueue(x_local)
        return self.finish()

    def gather_kept(self, res: ShardResult, dst: Optional[int] = None):
        """All ranks' kept rows in frame order (SURVEY.md §8e "output": the consumer is usually one rank's LLM
        prefill).  Two all-gathers: the per-rank counts, then the rows padded to the largest count (an all-gather-v).
        Returns (rows [K_total, D], global_idx [K_total], counts) on every rank, or only on rank `dst` (None
        elsewhere).  The kept rows stay sharded unless this is called."""
        if res.rows is None:
            raise RuntimeError("gather_kept needs a compressor built with gather=True")
        if self.world == 1:
            return res.rows, res.global_idx, [res.K]
        dev = res.rows.device
        counts = _all_gather(torch.tensor([res.K], dtype=torch.int64, device=dev), self.group, self.world).view(-1)
        counts = [int(c) for c in counts.tolist()]                  # (host sync: sizes of the result)
        kmax = max(counts)
        D = res.rows.shape[1]
        pad_rows = torch.zeros((kmax, D), dtype=res.rows.dtype, device=dev)
        pad_idx = torch.zeros(kmax, dtype=torch.int64, device=dev)
        pad_rows[: res.K] = res.rows
        pad_idx[: res.K] = res.global_idx
        all_rows = _all_gather(pad_rows, self.group, self.world)
        all_idx = _all_gather(pad_idx, self.group, self.world)
        if dst is not None and self.rank != dst:
            return None
        rows = torch.cat([all_rows[r, :c] for r, c in enumerate(counts)])
        gidx = torch.cat([all_idx[r, :c] for r, c in enumerate(counts)])
        return rows, gidx, counts
