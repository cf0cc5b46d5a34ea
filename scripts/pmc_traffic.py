#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes into profiles/<tag>_pmc_traffic.json (HBM bytes per kernel launch).

On the MI355X box (counters in separate runs, kernel-trace only -- see MI355X_MICROARCH.md §HBM / rocprofv3):

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- \
        python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- \
        python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra
    python scripts/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv r01_d

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  gfx950 correction (guide): FETCH_SIZE is exactly half of the
bytes of a wide coalesced streaming read (16 B / lane), so it is doubled for the streaming kernels; everything else
is left uncorrected (uncalibrated, small).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

STREAMING = {"k_chan_stats", "k_norm_colsum", "k_dist", "k_gather_rows"}


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"\b(k_[a-z_]+)", r["Kernel_Name"])
            if m:
                acc[m.group(1)].append(float(r["Counter_Value"]) * 1024.0)
    # the first dispatches include warm-up of other shapes (parity gate): keep the most common grid = median
    return {k: sorted(v)[len(v) // 2] for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    f_csv, w_csv, tag = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "target: 128x196x3584 bf16 r=0.25 (default 'torch' mode)"
    fetch, nf = per_kernel(f_csv, "FETCH_SIZE")
    write, _ = per_kernel(w_csv, "WRITE_SIZE")
    out = {"workload": workload,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB per dispatch, median over the "
                   "dispatches of the run). gfx950 correction per MI355X_MICROARCH.md HBM section: FETCH_SIZE reports "
                   "half of the bytes of a wide coalesced streaming read -> doubled for the streaming kernels "
                   "(k_chan_stats, k_norm_colsum, k_dist, k_gather_rows); WRITE_SIZE and the small kernels uncorrected.",
           "kernels": {}}
    for k in sorted(fetch):
        corr = 2.0 if k in STREAMING else 1.0
        out["kernels"][k] = {"fetch_bytes_raw": int(fetch[k]), "write_bytes": int(write.get(k, 0)),
                             "fetch_correction": corr, "hbm_bytes": int(fetch[k] * corr + write.get(k, 0)),
                             "dispatches": nf[k]}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", f"{tag}_pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)
    for k, v in out["kernels"].items():
        print(f"  {k:18s} {v['hbm_bytes'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
