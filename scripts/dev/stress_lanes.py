"""Clips in flight (compress_batch, 3-4 lanes) against the same clips one at a time: kept rows / indices must be equal -- the
selection workgroup of k_var_select POLLS for its variances, here next to other clips' sweeps.  python scripts/dev/stress_lanes.py"""
import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vidcom2_amd as vc
from vidcom2_amd import synth
dev = torch.device('cuda:0')
F, N, D = 128, 196, 3584
clips = [synth.make(F, N, D, torch.bfloat16, s, "drift").to(dev) for s in range(4)]
want = [vc.compress(c, N, 0.25) for c in clips]
torch.cuda.synchronize()
bad = 0
t0 = time.time()
for it in range(40):
    batch = [clips[(it + j) % 4] for j in range(12)]
    res = vc.vidcom2.compress_batch(batch, N, 0.25, in_flight=4)
    for j, r in enumerate(res):
        w = want[(it + j) % 4]
        if r.K != w.K or not torch.equal(r.global_idx, w.global_idx) or not torch.equal(r.rows, w.rows):
            bad += 1
print(f"{40 * 12} clips through compress_batch(in_flight=4): {bad} mismatches, {time.time() - t0:.1f}s")
# fp16 too
clips = [synth.make(F, N, 4096, torch.float16, s, "drift").to(dev) for s in range(3)]
want = [vc.compress(c, N, 0.25) for c in clips]
bad = 0
for it in range(20):
    batch = [clips[(it + j) % 3] for j in range(9)]
    res = vc.vidcom2.compress_batch(batch, N, 0.25, in_flight=3)
    for j, r in enumerate(res):
        w = want[(it + j) % 3]
        if r.K != w.K or not torch.equal(r.global_idx, w.global_idx):
            bad += 1
print(f"{20 * 9} fp16 clips through compress_batch(in_flight=3): {bad} mismatches")
