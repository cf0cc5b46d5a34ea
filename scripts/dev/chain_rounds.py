#!/usr/bin/env python
"""Per-ROUND cost of the two selection replays of one pass (library built with -DVC2_DEBUG_TIMING; VC2_ROUND stamps:
a counter in the caller's Sel2 and one store per stamp, no atomics):

    python scripts/dev/chain_rounds.py lib_dbg.so out.csv [F N D dtype]...

k_chan_select (slot 0) and frame 0 of k_select (slot 1).  One CSV row per stamp: kernel, case, stamp number, tag, what,
range length, us since the kernel's first stamp, us until the next stamp (= the cost of the step that BEGINS here), taken as the
MEDIAN over `reps` passes of the same input (the rounds are a deterministic function of the input)."""
import ctypes, os, statistics, sys
os.environ["VC2_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth

WHAT = {200: "load the variances (k_var_select: wait for the variance workgroups of the same launch)", 201: "pack words into LDS + barrier", 202: "(dispatch into the replay)",
        210: "cooperative round (16 waves; LDS)", 230: "one-wave round (LDS)", 240: "insertion sort (<= 3 elements)",
        250: "one-wave round (registers, <= 64 elements: introselect_tail64)", 310: "round on four waves, ballot form (sel3_rounds)",
        330: "one-wave round, ballot form (sel3_rounds)", 410: "round on the awake waves of 16 (sel4_round: thread-contiguous registers)",
        430: "round on ONE wave, no barrier (sel4_round<.., true>)", 290: "(return from the replay)", 291: "perm store; kept flags; block scan",
        292: "cols / mask stores; words for the ORDER riders", 299: "end",
        800: "load words + score partials; derive budgets", 801: "publish k / offsets; barrier", 802: "(dispatch into the replay)",
        803: "kept flags; scan; ordered compaction + index map", 809: "end"}

def cases(argv):
    if not argv:
        return [(128, 196, 3584, "bf16"), (128, 196, 3584, "f16"), (32, 196, 3584, "bf16"), (64, 324, 3584, "bf16")]
    out = []
    for i in range(0, len(argv), 4):
        out.append((int(argv[i]), int(argv[i + 1]), int(argv[i + 2]), argv[i + 3]))
    return out

def main():
    out_csv = sys.argv[2]
    L = ctypes.CDLL(_ffi.LIB_PATH)
    rows = ["kernel,case,stamp,tag,step_that_begins_here,range_len,us_from_first,us_until_next"]
    reps = 9
    for (F, N, D, dn) in cases(sys.argv[3:]):
        dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dn]
        base = 0.125 if N == 324 else 0.25
        x = synth.make(F, N, D, dt, 0, "drift").cuda()
        plan = vc.vidcom2.CompressPlan(F, N, D, dt, x.device, base)
        runs = {0: [], 1: []}
        for it in range(reps + 2):
            plan.enqueue(x); plan.finish()
            t = (ctypes.c_ulonglong * 256)(); v = (ctypes.c_int * 256)()
            L.vc2_debug_read_rounds(t, v)
            if it < 2:
                continue
            for slot in (0, 1):
                n = v[slot * 128 + 127]
                st = [(t[slot * 128 + i], v[slot * 128 + i] & 0xFFF, v[slot * 128 + i] >> 12) for i in range(n)]
                extra = [(t[slot * 128 + 120], v[slot * 128 + 120], 0)]
                if slot == 1:
                    extra.append((t[slot * 128 + 121], v[slot * 128 + 121], 0))
                runs[slot].append(extra + st)
        for slot, kname in ((0, "k_chan_select"), (1, "k_select[frame 0]")):
            rr = runs[slot]
            shape = [(s[1], s[2]) for s in rr[0]]
            same = [r for r in rr if [(s[1], s[2]) for s in r] == shape]      # (identical inputs: identical rounds)
            case = f"{F}x{N}x{D} {dn}"
            for i, (tag, ln) in enumerate(shape):
                rel = statistics.median((r[i][0] - r[0][0]) / 100.0 for r in same)
                step = statistics.median(((r[i + 1][0] - r[i][0]) / 100.0 if i + 1 < len(shape) else 0.0) for r in same)
                rows.append(f"{kname},{case},{i},{tag},{WHAT.get(tag, '?')},{ln},{rel:.2f},{step:.2f}")
    open(out_csv, "w").write("\n".join(rows) + "\n")
    print("\n".join(rows))

if __name__ == "__main__":
    main()
