"""compress_batch throughput vs clips in flight: python scripts/inflight.py <F> <N> <D> <f16|bf16> <clips>"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vidcom2_amd as vc
from vidcom2_amd import synth
F, N, D, dn, nclips = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[dn]
base = [synth.make(F, N, D, dt, s, "drift").cuda() for s in range(min(4, nclips))]
clips = [base[i % len(base)] for i in range(nclips)]
for k in (1, 2, 3, 4):
    for _ in range(3):
        vc.vidcom2.compress_batch(clips, N, 0.25, in_flight=k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        vc.vidcom2.compress_batch(clips, N, 0.25, in_flight=k)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / reps
    print(f"in_flight={k}: {dt_s * 1e3:.3f} ms per batch of {nclips} = {dt_s / nclips * 1e6:.1f} us per clip, {nclips * F * N / dt_s / 1e6:.1f} M tokens/s", flush=True)
