"""Torch-CPU op-for-op restatement of the compression pass (TEST / BASELINE INFRASTRUCTURE ONLY).

Why it exists: SURVEY.md §8d asks for "the reference CPU path" timed beside the GPU number, and the reference's own
.py cannot travel to the GPU box.  This file states the same pipeline with the same aten ops in the same order --
so that torch's CPU kernels do the same work with the same dtype behaviour as token_compressor/vidcom2/vidcom2.py
(file:line cited per step) -- written from the algorithm description in SURVEY.md §2 "op-level inventory", not copied
from the reference.  tests/test_oracle_golden.py pins it against the fixtures the reference produced (kept indices and
budgets equal on every case, bit-equal score digests), and bench.py times it as cpu_baseline kind "torch-restatement".
Only tests/, bench.py's cpu_baseline leg and tools may import it; the product never does.
"""
from __future__ import annotations

from typing import Dict, List

import torch

ALPHAS = [2.0 ** e for e in range(-3, 2)]                      # vidcom2.py:54


def low_variance_channels(x: torch.Tensor, ratio: float = 0.5):
    """vidcom2.py:38-43: per-channel population variance, the int(D*ratio) smallest, torch.topk's own order."""
    var = x.var(dim=0, unbiased=False)
    k = int(x.shape[-1] * ratio)
    idx = torch.topk(var, k, largest=False).indices
    return x[:, idx], idx


def _kernel_sum(z: torch.Tensor, centre: torch.Tensor) -> torch.Tensor:
    """vidcom2.py:59-62: squared distance to the centre, then the five Gaussian kernels added left to right
    starting from Python's integer 0."""
    d2 = ((z - centre) ** 2).sum(-1)
    total = 0
    for a in ALPHAS:
        total = total + torch.exp(-d2 / (2 * a))
    return total


def gaussian_scores(sel: torch.Tensor, tpf: int):
    """vidcom2.py:45-57: unit-normalised tokens against the video centre and their frame's centre."""
    z = torch.nn.functional.normalize(sel.view(-1, tpf, sel.shape[-1]), dim=-1)
    return _kernel_sum(z, z.mean(dim=(0, 1), keepdim=True)), _kernel_sum(z, z.mean(dim=1, keepdim=True))


def frame_scales(frame_scores: torch.Tensor, base: float, temp: float = 0.01) -> torch.Tensor:
    """vidcom2.py:64-68."""
    p = torch.softmax((frame_scores - frame_scores.max()) / temp, dim=0)
    return (base * (1 + p - p.mean())).clamp(max=1.0)


def kept_per_frame(total: torch.Tensor, scales: torch.Tensor, tpf: int) -> List[torch.Tensor]:
    """vidcom2.py:70-78: per frame the k_f smallest total scores, indices ascending."""
    ks = (scales * tpf).round().long().clamp(min=1).tolist()
    return [torch.topk(row, k, largest=False, sorted=False).indices.sort().values for row, k in zip(total, ks)], ks


def compress(x: torch.Tensor, tpf: int, base: float = 0.25) -> Dict[str, object]:
    """vidcom2.py:15-36 with the "linear" mapper: returns kept rows, global indices, budgets and the scores."""
    sel, chan = low_variance_channels(x)
    v, f = gaussian_scores(sel, tpf)
    scales = frame_scales(-v.mean(dim=-1), base)
    per_frame, ks = kept_per_frame(v + f, scales, tpf)
    gidx = torch.cat([idx + i * tpf for i, idx in enumerate(per_frame)])      # vidcom2.py:99-103
    return dict(rows=x[gidx], global_idx=gidx, ks=ks, v=v, f=f, chan_idx=chan)
