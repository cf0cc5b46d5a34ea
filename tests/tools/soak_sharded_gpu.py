"""GPU-box soak of the frame-sharded pass: random shapes and world sizes (emulated ranks on one GPU through the stage
entry points), token-by-token scores against the unsharded pass, in the default mode and in debug mode 2 (every
video-centre column through exchange 2b)."""
import os, sys, time, random, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vidcom2_amd import synth, _ffi
import vidcom2_amd as vc
from test_sharded import _emulate_ranks_on_one_gpu
lo, hi = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
bad = n = 0
t0 = time.time()
for seed in range(lo, hi):
    rng = random.Random(seed)
    P = rng.choice([2, 3, 4, 5, 6, 8])
    Fl = rng.choice([1, 2, 3, 4, 6, 8, 16])
    N = rng.choice([16, 37, 49, 100, 144, 169, 196, 324])
    D = rng.choice([64, 128, 256, 512, 1024, 3584])
    dt = rng.choice([torch.float16, torch.bfloat16])
    dist = rng.choice(["drift", "iid"])
    mode2 = rng.random() < 0.5 and D <= 1024
    F = P * Fl
    x = synth.make(F, N, D, dt, seed, dist).to(dev)
    try:
        if mode2:
            _ffi.lib().vc2_set_mode(2)
        else:
            _ffi.set_mode("torch")
        whole = vc.compress(x, N, 0.25, want_scores=True)
        total = (whole.v_score + whole.f_score).float().flatten()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res, st = _emulate_ranks_on_one_gpu(x, F, N, D, dt, 0.25, P, dev, vc_cap=max(64, D // 2))
        replayed = all(s.vc_fragile == 0 for s in st)
        same_idx = torch.equal(torch.cat([r.global_idx for r in res]), whole.global_idx)
        same_tot = torch.equal(torch.cat([s.total for s in st]), total)
        ok = same_idx and (same_tot or not replayed)
        if Fl * N >= 64 and not replayed:
            ok = False                                   # every case with a block's worth of rows per rank must replay
    finally:
        _ffi.set_mode("torch")
    n += 1
    if not ok:
        bad += 1
        print("MISMATCH", seed, P, Fl, N, D, dt, dist, "mode2" if mode2 else "torch", "replayed", replayed, same_idx, same_tot, flush=True)
print(f"{n} sharded cases, {bad} mismatches, {time.time() - t0:.0f}s")
