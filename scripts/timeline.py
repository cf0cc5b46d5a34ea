#!/usr/bin/env python
"""One-pass timeline from a rocprofv3 kernel trace: python scripts/timeline.py <kernel_trace.csv> [out.csv]
Picks the LAST complete pass (k_chan_stats ... k_gather_rows) and prints start / duration / gap per kernel."""
import csv, re, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        m = re.search(r"\b(k_[a-z_0-9]+)", r["Kernel_Name"])
        if m:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Queue_Id", "")))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2] == "k_gather_rows"]
starts = [i for i, r in enumerate(rows) if r[2] == "k_chan_stats"]
e = ends[-2] if len(ends) > 1 else ends[-1]
s = max(i for i in starts if i < e)
t0 = rows[s][0]
out = ["kernel,queue,start_us,dur_us,gap_before_us"]
prev_end = t0
for a, b, name, q in rows[s:e + 1]:
    out.append(f"{name},{q},{(a - t0) / 1e3:.1f},{(b - a) / 1e3:.1f},{(a - prev_end) / 1e3:.1f}")
    prev_end = max(prev_end, b)
out.append(f"TOTAL,,{0:.1f},{(rows[e][1] - t0) / 1e3:.1f},")
print("\n".join(out))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(out) + "\n")
