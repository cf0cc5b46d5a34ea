"""Efficiency report of an evaluation run.

The reference's lmms-eval model wrappers time every `model.generate` call with device events, track the peak
allocated memory and print, on rank 0 at the end of the run, a three-row table -- `LLM_time_s`, `Total_time_s`,
`Peak_mem_MB` (lmms-eval/lmms_eval/models/llava_onevision.py:595-634, table layout :65-77).  `EfficiencyMeter` is that
bookkeeping as a reusable object for a ROCm run (torch.cuda.Event is a hipEvent there):

    meter = EfficiencyMeter()                      # starts the wall clock
    for request in requests:
        with meter.generation():                   # sync, reset peak stats, record start ... record end, sync
            out = model.generate(...)
    print(meter.table())                           # rank 0

`stage("Compress")` times additional device stages with their own event pairs (e.g. the compression pass inside the
prefill) and adds a `<name>_time_s` row; the three reference rows always come first and keep their formats.
"""
from __future__ import annotations

import contextlib
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch

__all__ = ["EfficiencyMeter", "format_efficiency_table"]


def format_efficiency_table(rows: Sequence[Tuple[str, str]], title: str = "Efficiency Analysis") -> str:
    """ASCII table `| Metric | Value |` with left-aligned cells, a rule above and below the header and below the
    last row, preceded by the title line (the reference's layout)."""
    rows = [(str(m), str(v)) for m, v in rows]
    head = ("Metric", "Value")
    w0 = max([len(head[0])] + [len(m) for m, _ in rows])
    w1 = max([len(head[1])] + [len(v) for _, v in rows])
    rule = "+" + "-" * (w0 + 2) + "+" + "-" * (w1 + 2) + "+"

    def line(a: str, b: str) -> str:
        return "| " + a.ljust(w0) + " | " + b.ljust(w1) + " |"

    out: List[str] = [title, rule, line(*head), rule]
    out += [line(m, v) for m, v in rows]
    out.append(rule)
    return "\n".join(out)


class EfficiencyMeter:
    def __init__(self, device: Optional[torch.device] = None):
        self.device = device
        self.total_cuda_time = 0.0          # seconds inside generation(), by device events
        self.max_mem = 0.0                  # MB, peak torch allocation inside any generation()
        self.calls = 0
        self._stages: Dict[str, float] = {}
        self._wall_start = time.time()

    def restart_wall_clock(self) -> None:
        self._wall_start = time.time()

    def _require_gpu(self) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("vidcom2_amd.efficiency: device events need a ROCm device (no CPU fallback)")

    @contextlib.contextmanager
    def generation(self):
        """One `model.generate` call: event-timed, with the peak-memory watermark reset before it."""
        self._require_gpu()
        dev = self.device
        torch.cuda.reset_peak_memory_stats(dev)
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        start.record()
        try:
            yield self
        finally:
            end.record()
            torch.cuda.synchronize(dev)
            self.total_cuda_time += start.elapsed_time(end) / 1000.0
            self.max_mem = max(self.max_mem, torch.cuda.max_memory_allocated(dev) / 1024 / 1024)
            self.calls += 1

    @contextlib.contextmanager
    def stage(self, name: str):
        """An extra device stage (its own event pair on the current stream); adds the row `<name>_time_s`."""
        self._require_gpu()
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        try:
            yield self
        finally:
            end.record()
            end.synchronize()
            self._stages[name] = self._stages.get(name, 0.0) + start.elapsed_time(end) / 1000.0

    def rows(self, wall_time: Optional[float] = None) -> List[Tuple[str, str]]:
        wall = time.time() - self._wall_start if wall_time is None else wall_time
        rows = [("LLM_time_s", f"{self.total_cuda_time:.3f}"),
                ("Total_time_s", f"{wall:.3f}"),
                ("Peak_mem_MB", f"{self.max_mem:.1f}")]
        rows += [(f"{k}_time_s", f"{v:.3f}") for k, v in self._stages.items()]
        return rows

    def table(self, wall_time: Optional[float] = None, title: str = "Efficiency Analysis") -> str:
        return format_efficiency_table(self.rows(wall_time), title)
