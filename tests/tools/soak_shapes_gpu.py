"""GPU-box soak over SHAPES: random frame counts, tokens per frame, widths (vector and scalar paths), dtypes and
retain ratios against the oracle -- scores, budgets, kept indices (torch mode)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle as O
from vidcom2_amd import synth, _ffi
from vidcom2_amd.vidcom2 import compress
PRIMARY = os.environ.get("VC2_SOAK_MODE", "torch")       # the mode under test (the default; "torch_fast" shows the cancel misses of rounds 1-3)
O.set_mode("torch"); _ffi.set_mode(PRIMARY)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
LARGE = os.environ.get("VC2_SOAK_LARGE", "0") != "0"      # shapes of 6e7 .. 2.2e8 elements (target = 9e7), which the default run skips
ONLY_DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32, "": None}[os.environ.get("VC2_SOAK_DTYPE", "")]
bad = n = bad3 = 0
t0 = time.time()
for seed in range(lo, hi):
    rng = random.Random(seed)
    F = rng.choice([1, 2, 3, 5, 8, 13, 31, 64, 100, 257, 600])
    N = rng.choice([1, 2, 3, 7, 16, 17, 49, 100, 169, 196, 255, 400])
    D = rng.choice([8, 24, 64, 72, 200, 256, 520, 1000, 1024, 2048, 3584, 4096])
    if LARGE:                                 # (round 5, VERDICT r4 item 4: large shapes only, half precision, mostly `cancel`)
        F = rng.choice([64, 100, 128, 160, 257])
        N = rng.choice([169, 196, 255, 324, 400])
        D = rng.choice([1024, 2048, 3584, 4096])
        if not (6e7 < F * N * D <= 2.2e8):
            continue
    elif F * N * D > 6e7:
        continue
    dt = rng.choice([torch.float16, torch.bfloat16, torch.float32])
    dist = rng.choice(["drift", "iid", "cancel"])
    if LARGE:
        dt = rng.choice([torch.float16, torch.bfloat16])
        dist = rng.choice(["cancel", "cancel", "drift", "iid"])
    if ONLY_DT is not None:                   # (VC2_SOAK_DTYPE=f16|bf16|f32: one dtype only)
        dt = ONLY_DT
    base = rng.choice([0.05, 0.15, 0.25, 0.5, 0.9])
    x = synth.make(F, N, D, dt, seed, dist)
    try:
        r = compress(x.cuda(), N, base, want_scores=True)
    except NotImplementedError:
        continue
    if dt == torch.float32:
        # fp32 inputs: the HIP path uses correctly rounded (fp64) reductions whatever the mode -- torch's fp32 accumulation
        # order is not replayed for fp32 (scores within 1e-5 of the reference, near-tied kept
        # indices may flip) -- so the like-for-like check is against the oracle's 'exact' mode
        O.set_mode("exact")
        o = O.compress_indices(x, N, base)
        O.set_mode("torch")
        ok = torch.equal(r.global_idx.cpu(), o["global_idx"]) and torch.equal(r.ks.cpu(), o["ks"]) and \
            torch.allclose(r.v_score.cpu(), o["v"], atol=1e-5, rtol=0, equal_nan=True)
    else:
        o = O.compress_indices(x, N, base)
        eq = lambda a, b: torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))
        ok = (torch.equal(r.global_idx.cpu(), o["global_idx"]) and eq(r.v_score.cpu(), o["v"])
              and eq(r.f_score.cpu(), o["f"]) and torch.equal(r.ks.cpu(), o["ks"]))
    n += 1
    if n % (50 if LARGE else 1000) == 0:                        # (progress: a cut-off run still says how far it got)
        print(f"[mode {PRIMARY}] seeds {lo}..{seed}: {n} cases, {bad} mismatches so far, {time.time() - t0:.0f}s", flush=True)
    if not ok:
        bad += 1
        proven = None
        if dt != torch.float32:              # the robust mode must get it (`cancel` inputs)
            _ffi.set_mode(os.environ.get("VC2_SOAK_FALLBACK_MODE", "torch_proven"))
            r3 = compress(x.cuda(), N, base, want_scores=True)
            _ffi.set_mode(PRIMARY)
            proven = (torch.equal(r3.global_idx.cpu(), o["global_idx"]) and eq(r3.v_score.cpu(), o["v"])
                      and eq(r3.f_score.cpu(), o["f"]) and torch.equal(r3.ks.cpu(), o["ks"]))
            bad3 += 0 if proven else 1
        print("MISMATCH", seed, F, N, D, dt, dist, base, "| robust mode matches:", proven, flush=True)
print(f"[mode {PRIMARY}] {n} shape cases, {bad} mismatches in this mode ({bad3} of them also in the robust mode), {time.time() - t0:.0f}s")
