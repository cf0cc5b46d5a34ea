#!/usr/bin/env python
"""Average / min duration per kernel from a rocprofv3 kernel_stats csv: python scripts/kstats.py <dir or csv> [...]"""
import csv, glob, os, re, sys
for a in sys.argv[1:]:
    f = a if a.endswith(".csv") else (glob.glob(a + "/**/*kernel_stats.csv", recursive=True) or [None])[0]
    if not f:
        continue
    rows = []
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(k_[a-z_0-9]+)", r["Name"])
        if m:
            name = {"k_norm_colsum2": "k_norm_colsum", "k_dist2": "k_dist", "k_chan_select3": "k_chan_select", "k_chan_select4": "k_chan_select", "k_var_select": "k_chan_select"}.get(m.group(1), m.group(1))   # (the streamlined sweeps)
            rows.append((name, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
    order = ["k_chan_stats", "k_var_from_stats", "k_chan_select", "k_norm_colsum", "k_norm_fix", "k_frame_centres",
             "k_video_centre", "k_dist", "k_select", "k_gather_rows"]
    rows.sort(key=lambda r: order.index(r[0]) if r[0] in order else 99)
    tot = sum(r[2] for r in rows if r[0] in order)
    print(os.path.relpath(f), "| pass sum of averages %.1f us" % tot)
    print("   " + "  ".join(f"{n[2:]} {a:.1f}" for n, c, a, mn in rows if n in order))
