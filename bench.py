#!/usr/bin/env python
"""bench.py -- throughput of the VidCom2 token-compression pass on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload target|cfg4|cfg5|...]
      N > 1 without WORLD_SIZE in the environment: the script re-launches itself under torch.distributed.run with N
      ranks on 127.0.0.1 (one rank per GPU; fails loudly when the node has fewer than N devices).  Under a launcher
      (WORLD_SIZE set) --gpus must equal WORLD_SIZE.

A "step" = one whole pass over one batch of synthetic frame-token embeddings already resident in
HBM: channel selection, scoring, budgets, per-frame selection and the kept-row gather
(vidcom2.py:15-36).  Metric (BASELINE.json): input video tokens compressed per second at the
stated retain ratio, with kept indices / budgets bit-exact vs the reference oracle (checked here on
every run before timing).

N = 1 : workload "target" = 128 frames x 196 tokens x 3584-d bf16, 25 % retain (the shape
        BASELINE.json's north-star target is quoted on); cfg2 (32x196x3584) is reported beside it.
N > 1 : "target": weak scaling -- every rank holds 128 frames of ONE long video of 128*N frames, frame-sharded
        with three (+1) small RCCL all-gathers (stat blocks, centre sums, replay blocks, per-frame uniqueness scores);
        "cfg4" (BASELINE configs[3]): STRONG scaling -- one 512-frame video, 512/N frames per rank, same exchanges;
        "cfg5" (configs[4]): 16 clips of 128x196x4096 fp16, 16/N clips per rank, replicas (no collective: the
        reference's document-level data parallelism), three clips in flight per GPU (compress_batch's default).
        The sharded lines carry "exchanges_us": the mean time of every all-gather with its message size.

Environment knobs: VC2_BENCH_WATCHDOG=<s> (a run still going after s seconds dumps every thread's Python stack and
exits), VC2_BENCH_COLL_TIMEOUT=<s> (collective timeout, default 600), VC2_BENCH_CPU_THREADS="16,32" (thread counts of
the CPU-baseline sweep), VC2_BENCH_FORCE_DIST=1 (the sharded path and its collectives at world size 1),
VC2_BENCH_ONE_GPU=1 + VC2_BENCH_BACKEND=gloo (tests: every rank on cuda:0).

One JSON line on rank 0 (see the repo prompt for the contract), with extra objects:
  "roofline"      the dominant kernel's achieved algorithmic HBM rate (hipEvent-timed inside this run)
  "roofline_big"  the same for a > 256 MiB working set (cfg3), where the Infinity Cache cannot hold X
  "cpu_baseline"  the reference CPU path on this box's host cores: kind "port", flavour "torch-restatement" (the same
                  aten ops, oracle/torch_restatement.py) and, nested under "port", the C++/OpenMP oracle -- each at the
                  best of several thread counts (all 256 hardware threads oversubscribe torch's small ops;
                  VC2_BENCH_CPU_THREADS="16,32" restricts the sweep -- the test-suite's subprocess runs do)
  "pass_roofline" the whole pass against 8 TB/s in three states of the input: warm (= value), cold, producer_warm
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
MALL_BYTES = 256 << 20    # Infinity Cache

WORKLOADS = {
    # name: (F, N, D, dtype, base_scale)
    "target": (128, 196, 3584, torch.bfloat16, 0.25),
    "cfg2": (32, 196, 3584, torch.bfloat16, 0.25),
    "cfg3": (64, 324, 3584, torch.bfloat16, 0.125),
    "target_fp32": (128, 196, 3584, torch.float32, 0.25),
    "cfg5clip": (128, 196, 4096, torch.float16, 0.25),
    "target_f16": (128, 196, 3584, torch.float16, 0.25),   # the target shape in the reference's LLaVA dtype (llava_onevision.py:511)
    "cfg1": (8, 196, 1024, torch.float32, 0.25),
    "cfg4": (512, 196, 3584, torch.bfloat16, 0.25),       # one long video; frame-sharded: 512 / world frames per rank
    "cfg5": (128, 196, 4096, torch.float16, 0.25),        # per CLIP; the workload is 16 of them, 16 / world per rank
}
CFG5_CLIPS = 16
DT_NAME = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}


def alg_bytes_pass(F, N, D, es, r):
    """three sweeps of X + gather (read+write of r*X) + score/index traffic."""
    return F * N * D * es * (3 + 2 * r) + 12 * F * N + 8 * r * F * N


def kernel_alg_bytes(name, F, N, D, es, K):
    """Algorithmic HBM bytes of ONE launch of a kernel (DESIGN.md 'Kernels')."""
    X = F * N * D * es
    return {
        "k_chan_stats": X,                       # sweep 1: read X once
        "k_norm_colsum": X + 4 * F * N,          # sweep 2: read X, write den
        "k_dist": X + 12 * F * N,                # sweep 3: read X + den, write scores
        "k_gather_rows": 2 * K * D * es + 8 * K,  # read K rows + indices, write K rows
    }.get(name)


def pmc_traffic(kernel, workload, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command, gfx950 FETCH_SIZE x2
    correction applied as MI355X_MICROARCH.md prescribes).  None if no profile matches this workload."""
    import glob
    if world != 1:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("workload", "").startswith(workload + ":") and kernel in d.get("kernels", {}):
            return d["kernels"][kernel]["hbm_bytes"]
    return None


def time_steps(fn, steps, dist_on):
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier(device_ids=[torch.cuda.current_device()])
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier(device_ids=[torch.cuda.current_device()])
        torch.cuda.synchronize()
    return time.perf_counter() - t0


def warm_clocks(step, multi_rank, sync=None, seconds=0.3, fixed_rounds=5):
    """Passes that bring the GPU clocks back up before the warm-up and timed steps; returns how many were issued.
    One rank: about `seconds` of them.  More than one rank: a CONSTANT number -- every pass of the frame-sharded path
    issues collectives, and how many a rank issues must not depend on that rank's own clock (tests/test_abi_and_host.py)."""
    sync = sync or torch.cuda.synchronize
    t0 = time.perf_counter()
    rounds = 0
    while (rounds < fixed_rounds) if multi_rank else (time.perf_counter() - t0 < seconds):
        for _ in range(10):
            step()
        sync()
        rounds += 1
    return rounds * 10


def median_step_ms(fn, steps):
    """Median over `steps` individually event-timed passes (SURVEY.md §8d asks for the median; the contract's
    ms_per_step above is the mean of the K-step block)."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)


def cpu_baselines(x_cpu, N, base, budget_s=10.0):
    """The reference CPU path on this box's host cores, two ways, on full passes of the same workload:
    'torch-restatement' -- the same aten ops in the same order (oracle/torch_restatement.py, pinned bit-exact against
    the reference's fixtures) with torch's own threading; 'port' -- the C++/OpenMP oracle (oracle/vc2_oracle.cpp).
    Both are run at several thread counts (one pass each) and then timed at the best one: with every hardware thread
    (256 here) torch's small ops oversubscribe and the same pass runs an order of magnitude slower than with 16-32."""
    import oracle
    from oracle import torch_restatement as T
    cores = os.cpu_count() or 1
    counts = sorted({c for c in (8, 16, 32, 64, 128, 256, cores) if c <= cores})
    if os.environ.get("VC2_BENCH_CPU_THREADS"):         # (the test-suite's subprocess runs: the parity gate, not the sweep)
        counts = sorted({min(cores, int(c)) for c in os.environ["VC2_BENCH_CPU_THREADS"].split(",")})

    def best_of(run, set_threads):
        tried = {}
        for c in counts:
            set_threads(c)
            run(x_cpu[: 8 * N])                         # warm at this count (thread pool, page-in)
            ts = []
            for _ in range(2):                          # (the second pass: caches and the thread pool are warm)
                t0 = time.perf_counter()
                run(x_cpu)
                ts.append(time.perf_counter() - t0)
            tried[c] = min(ts)
            if tried[c] > 4 * min(tried.values()):      # far off the best already: larger counts only get worse
                break
        c = min(tried, key=tried.get)
        set_threads(c)
        t0 = time.perf_counter()
        reps, last = 0, None
        while True:
            last = run(x_cpu)
            reps += 1
            if time.perf_counter() - t0 > budget_s or reps >= 10:
                break
        return c, (time.perf_counter() - t0) / reps, reps, last, tried

    c1, dt, reps, rt, tried1 = best_of(lambda xx: T.compress(xx, N, base), torch.set_num_threads)
    out = {"value": x_cpu.shape[0] / dt, "unit": "tokens/s", "cores": c1, "kind": "port",
           "flavour": "torch-restatement",       # (oracle/torch_restatement.py: the reference's own torch CPU ops, op for op)
           "sample": f"{reps} full passes of the same workload ({x_cpu.shape[0]} tokens each), {dt * 1e3:.1f} ms/pass, "
                     f"torch CPU ops in the reference's order, {c1} threads = the best of "
                     + ", ".join(f"{c}: {t * 1e3:.0f} ms" for c, t in tried1.items()) + f" (host has {cores})"}
    oracle.set_mode("torch")             # same semantics as the HIP path's default mode

    def run_oracle(xx):
        o = oracle.compress_indices(xx, N, base)
        _ = xx[o["global_idx"]]
        return o
    c2, dt2, reps2, o, tried2 = best_of(run_oracle, oracle.set_num_threads)
    out["port"] = {"value": x_cpu.shape[0] / dt2, "unit": "tokens/s", "cores": c2, "kind": "port",
                   "sample": f"{reps2} full passes, {dt2 * 1e3:.1f} ms/pass, C++/OpenMP oracle in 'torch order' mode, {c2} threads "
                             "= the best of " + ", ".join(f"{c}: {t * 1e3:.0f} ms" for c, t in tried2.items())}
    torch.set_num_threads(cores)
    # the two CPU statements and the reference agree: same kept indices
    assert rt["ks"] == o["ks"].tolist() and torch.equal(rt["global_idx"], o["global_idx"]), "CPU baselines disagree"
    return out, o


def kernel_profile(step, steps, _ffi):
    """us per launch of every kernel, hipEvents around each launch (the ORDER replay runs as its own kernel here)."""
    _ffi.profile_enable(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    prof = _ffi.profile_collect()
    _ffi.profile_enable(False)
    return {name: round(tot / cnt * 1e3, 2) for name, (tot, cnt) in prof.items()}


def roofline_of(kern, F, N, D, es, K, workload, world):
    sweeps = {n: kern[n] for n in ("k_chan_stats", "k_norm_colsum", "k_dist", "k_gather_rows") if n in kern}
    if not sweeps:
        return None
    dom = max(sweeps, key=sweeps.get)
    ab = kernel_alg_bytes(dom, F, N, D, es, K)
    ach = ab / (sweeps[dom] * 1e-6) / 1e9
    X = F * N * D * es
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, workload, world),
            "alg_bytes_per_launch": ab, "avg_us": sweeps[dom],
            "x_bytes": X, "x_fits_infinity_cache": bool(X < MALL_BYTES),
            "note": ("X (%d MB) fits the 256 MiB Infinity Cache: sweeps 2-3 re-read it from there, so 'achieved' is an "
                     "algorithmic rate that may exceed what HBM alone delivers" % (X >> 20)) if X < MALL_BYTES else
                    ("X (%d MB) exceeds the 256 MiB Infinity Cache: every sweep streams from HBM" % (X >> 20))}


def relaunch_under_launcher(args) -> int:
    """`python bench.py --gpus N` (N > 1, no launcher): run the same command as N ranks of ONE node."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but this node exposes {have} GPU(s): refusing to report a "
                         f"{args.gpus}-GPU number from fewer devices")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (the host driver supports nothing else)
    return subprocess.call(cmd, env=env)


def bench_replicas(args, rank, world, dev, vc, _ffi, synth):
    """cfg5: 16 independent clips, 16 / world per rank, no collective (the reference's document-level DP,
    lmms-eval evaluator.py:488-491); every rank keeps compress_batch's default number of clips in flight (three), one stream each."""
    F, N, D, dtype, base = WORKLOADS["cfg5"]
    if CFG5_CLIPS % world:
        raise SystemExit(f"[bench] cfg5: {CFG5_CLIPS} clips do not split over {world} ranks")
    mine = CFG5_CLIPS // world
    # distinct clips (different seeds); generated on the host once, resident in HBM before anything is timed
    clips = [synth.make(F, N, D, dtype, seed=rank * mine + i, dist="drift").to(dev) for i in range(min(mine, 4))]
    clips = [clips[i % len(clips)] for i in range(mine)]        # (>4 clips per rank: the same four again -- timing only)
    res0 = vc.vidcom2.compress_batch(clips[:1], N, base)[0]
    if rank == 0 and not args.no_cpu_baseline:                  # parity gate on the first clip
        import oracle
        oracle.set_mode("torch")
        ref = oracle.compress_indices(clips[0].cpu(), N, base)
        if res0.ks.cpu().tolist() != ref["ks"].tolist() or not torch.equal(res0.global_idx.cpu(), ref["global_idx"]):
            raise SystemExit("[bench] PARITY FAILURE (cfg5 clip 0): kept indices / budgets differ from the oracle")
    step = lambda: vc.vidcom2.compress_batch(clips, N, base)      # noqa: E731
    for _ in range(args.warmup):
        step()
    dist_on = world > 1
    elapsed = time_steps(step, args.steps, dist_on)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3
    es = 2
    out = {
        "metric": "video-tokens compressed/sec at 25% retain; kept-index bit-exact vs ref",
        "value": round(CFG5_CLIPS * F * N / (elapsed / args.steps), 1), "unit": "tokens/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": DT_NAME[dtype], "data": "synthetic",
        "config": {"workload": f"cfg5: {CFG5_CLIPS} clips x {F} frames x {N} tokens x {D}-d {DT_NAME[dtype]}, retain {base}, "
                               f"{mine} clip(s) per GPU, two in flight, no collective (replicas)",
                   "kept_tokens_clip0": res0.K, "parallelism": f"replicas x{world}"},
        "roofline": None, "cpu_baseline": None,
        "pass_roofline": {"alg_bytes": alg_bytes_pass(F, N, D, es, base) * CFG5_CLIPS,
                          "achieved_GBs": round(alg_bytes_pass(F, N, D, es, base) * CFG5_CLIPS / (ms * 1e-3) / 1e9, 1),
                          "frac_of_8TBs_per_gpu": round(alg_bytes_pass(F, N, D, es, base) * mine / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "mode": _ffi.get_mode(),
    }
    if rank == 0:
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements")
    args = ap.parse_args()
    if os.environ.get("VC2_BENCH_WATCHDOG"):
        # a run that is still going after this many seconds dumps every thread's Python stack to stderr and exits (the
        # test-suite's subprocess runs set it: a hung rank says where instead of sitting out a 30-minute collective timeout)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["VC2_BENCH_WATCHDOG"]), exit=True)

    if args.gpus < 1:
        raise SystemExit("[bench] --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and os.environ.get("VC2_BENCH_FORCE_DIST") != "1":
        raise SystemExit(relaunch_under_launcher(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s): the line "
                         "would report a GPU count it did not run on")
    force = os.environ.get("VC2_BENCH_FORCE_DIST") == "1"       # the sharded code path + its collectives at world 1
    dist_on = world > 1 or force
    if os.environ.get("VC2_BENCH_ONE_GPU") == "1":    # test knob: every rank on cuda:0 (with VC2_BENCH_BACKEND=gloo;
        local = 0                                     # RCCL refuses two ranks on one device)
    elif torch.cuda.device_count() <= local:
        raise SystemExit(f"[bench] rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # lazy communicator creation on purpose: with `device_id=` (eager init) every pass measured ~45 us slower on
        # this stack in round 1; with it the numbers match a plain process
        # (a collective that has not completed after 10 minutes is an error, not a 30-minute wait (gloo): the longest
        #  legitimate wait is the other ranks' for rank 0's CPU parity gate, tens of seconds)
        import datetime
        torch.distributed.init_process_group(os.environ.get("VC2_BENCH_BACKEND", "nccl"),
                                             timeout=datetime.timedelta(seconds=int(os.environ.get("VC2_BENCH_COLL_TIMEOUT", "600"))))

    import vidcom2_amd as vc
    from vidcom2_amd import _ffi, synth

    if args.workload == "cfg5":
        bench_replicas(args, rank, world, dev, vc, _ffi, synth)
        if dist_on:
            torch.distributed.destroy_process_group()
        return
    F, N, D, dtype, base = WORKLOADS[args.workload]
    es = 4 if dtype == torch.float32 else 2
    strong = args.workload == "cfg4"                   # one video of F frames, F / world per rank
    if strong:
        if F % world or (F // world) % 8:
            raise SystemExit(f"[bench] cfg4: {F} frames do not split into multiples of 8 over {world} ranks")
        F_total, F = F, F // world
    else:
        F_total = F * world                            # weak scaling: every rank brings F frames

    # ---- synthetic input, resident in HBM before any timed region --------------------------------
    x_cpu = None
    if not dist_on:
        x_cpu = synth.make(F, N, D, dtype, seed=0, dist="drift")
        x = x_cpu.to(dev)
    else:
        # each rank generates only its own 128-frame shard of the 128*world-frame video
        x32 = synth.make_fp32_frames(F_total, N, D, rank * F, F, seed=0, dist="drift")
        x = synth.to_torch(x32, dtype).reshape(F * N, D).to(dev)
        del x32
        if force and world == 1:
            x_cpu = x.cpu()                            # (one rank: the CPU reference can check the sharded path too)

    if not dist_on:
        plan = vc.vidcom2.CompressPlan(F, N, D, dtype, dev, base)
        step = lambda: plan.enqueue(x)          # noqa: E731
        finish = plan.finish
    else:
        from vidcom2_amd.sharded import ShardedCompressor
        sc = ShardedCompressor(F, N, D, dtype, dev, base, group=None, always_collective=force)
        step = lambda: sc.enqueue(x)            # noqa: E731
        finish = sc.finish

    # ---- parity gate: kept indices + budgets must equal the CPU reference path's before anything is timed ------
    cpu = None
    step()
    res = finish()
    if x_cpu is not None and not args.no_cpu_baseline:
        cpu, ref = cpu_baselines(x_cpu, N, base, budget_s=10.0 if F_total * N <= 30000 else 3.0)
        ok = res.ks.cpu().tolist() == ref["ks"].tolist() and torch.equal(res.global_idx.cpu(), ref["global_idx"])
        if not ok:
            raise SystemExit("[bench] PARITY FAILURE: kept indices / budgets differ from the oracle")
    K = res.K

    # ---- timed region -------------------------------------------------------------------------------
    # (setup, not warm-up: the CPU parity gate above kept the GPU idle for ~20 s and it comes back clocked down; a third
    # of a second of passes brings the clocks up before the W warm-up steps and the K timed steps of the contract)
    # (More than one rank: a FIXED number of passes.  A loop bounded by each rank's own clock ran 30 passes on one rank
    #  and 40 on the other every so often -- mismatched all-gathers, both ranks hung until the 30-minute collective
    #  timeout: the intermittent hang of the two-rank tests, found with VC2_BENCH_WATCHDOG in round 4.)
    warm_clocks(step, world > 1)
    for _ in range(args.warmup):
        step()
    elapsed = time_steps(step, args.steps, dist_on)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3
    tokens_per_s = F_total * N / (elapsed / args.steps)

    # ---- roofline leg: same steps again with hipEvents around every kernel --------------------------
    kern = kernel_profile(step, args.steps, _ffi)
    roof = roofline_of(kern, F, N, D, es, K, args.workload, world)

    out = {
        "metric": "video-tokens compressed/sec at 25% retain; kept-index bit-exact vs ref",
        "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": DT_NAME[dtype], "data": "synthetic",
        "config": {"workload": f"{args.workload}: {F_total} frames x {N} tokens x {D}-d {DT_NAME[dtype]}, "
                               f"retain {base}" + (f", frame-sharded {F} frames/GPU, 4 RCCL all-gathers" if dist_on else ""),
                   "frames_per_gpu": F,
                   "kept_tokens": K, "parallelism": f"frame-shard x{world}" if dist_on else "single GPU"},
        "roofline": roof,
        "cpu_baseline": cpu,
        "pass_roofline": {"alg_bytes": alg_bytes_pass(F, N, D, es, base) * world,
                          "achieved_GBs": round(alg_bytes_pass(F, N, D, es, base) * world / (ms * 1e-3) / 1e9, 1),
                          "frac_of_8TBs_per_gpu": round(alg_bytes_pass(F, N, D, es, base) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "kernels_us": kern,
        "kernels_us_note": ("STAND-ALONE launches with hipEvents around each one (the ORDER replay and the norm fix-ups run as "
                            "their own kernels here; in a real pass they ride in the sweep-2 / centre launches), so the sum "
                            "exceeds ms_per_step; names vs rocprofv3: k_stats_reduce = k_var_from_stats, k_centres = "
                            "k_frame_centres + k_video_centre, k_norm_colsum = k_norm_colsum2 (the streamlined sweep 2)"),
    }
    extra_wanted = not dist_on and not args.no_extra
    out["mode"] = _ffi.get_mode() + (" (bit-exact to the CPU reference: boundary-fragile tokens replay torch's fp32 "
                                     "accumulation order)" if _ffi.get_mode() == "torch" else "")
    if not dist_on:
        # `value` / ms_per_step above is the WARM number: the loop re-reads one tensor, and X (171 MiB at the target
        # shape) is largely still in the 256 MiB Infinity Cache when the next step starts.  Two more states of the input,
        # same plan, same kernels, each with its fraction of the 8 TB/s roofline:
        #   cold            three different clips in turn (513 MiB of inputs: a step's X was evicted by the two before)
        #   producer_warm   X is WRITTEN by a device copy right before every pass, as the projector / pooling kernel
        #                   of the model would leave it (the copy is not timed: hipEvents around the pass only)
        ab = alg_bytes_pass(F, N, D, es, base)
        pr = out["pass_roofline"]
        pr["input_state"] = "warm (the same tensor every step)"
        nst = max(args.steps, 20)
        xs3 = [x] + [synth.make(F, N, D, dtype, seed=sd, dist="drift").to(dev) for sd in (1, 2)]
        turn = [0]

        def rotate():
            plan.enqueue(xs3[turn[0] % 3])
            turn[0] += 1
        for _ in range(max(args.warmup, 6)):
            rotate()
        e_cold = time_steps(rotate, nst, False) / nst
        pr["cold_ms_per_step"] = round(e_cold * 1e3, 4)
        pr["cold_frac_of_8TBs"] = round(ab / e_cold / 1e9 / HBM_PEAK_GBS, 4)
        pr["cold_tokens_per_s"] = round(F * N / e_cold, 1)
        xw = torch.empty_like(x)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nst)]
        for i in range(5):
            xw.copy_(xs3[i % 3]); plan.enqueue(xw)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(ev):
            xw.copy_(xs3[i % 3])
            a.record(); plan.enqueue(xw); b.record()
        torch.cuda.synchronize()
        e_pw = statistics.mean(a.elapsed_time(b) for a, b in ev) * 1e-3
        pr["producer_warm_ms_per_step"] = round(e_pw * 1e3, 4)
        pr["producer_warm_frac_of_8TBs"] = round(ab / e_pw / 1e9 / HBM_PEAK_GBS, 4)
        pr["note"] = ("value / ms_per_step = the warm state; cold = three clips in turn; producer_warm = X written by a "
                      "device copy immediately before each pass (hipEvents around the pass)")
        x_long = torch.cat([xs3[0], xs3[1], xs3[2], xs3[0]]) if (extra_wanted and args.workload == "target") else None
        del xs3, xw
    if not dist_on:
        out["median_ms_per_step"] = round(median_step_ms(step, max(args.steps, 20)), 4)
    else:
        # the exchanges of the sharded pass, one by one: hipEvents around every all-gather of `steps` more passes
        sc.profile_exchanges(True)
        for _ in range(max(5, args.steps // 2)):
            step()
        out["exchanges_us"] = sc.exchange_times_us()
        sc.profile_exchanges(False)

    extra = not dist_on and not args.no_extra
    # ---- side: the pass replayed from a hipGraph (serving loops capture it once) ------------------------
    if extra:
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                step()
            for _ in range(args.warmup):
                g.replay()
            eg = time_steps(g.replay, args.steps, False)
            out["hipgraph_replay"] = {"ms_per_step": round(eg / args.steps * 1e3, 4),
                                      "tokens_per_s": round(F * N / (eg / args.steps), 1)}
        except Exception as e:  # noqa: BLE001
            out["hipgraph_replay"] = {"error": repr(e)[:200]}
    # ---- side: one-shot call latency = the plugin API as the hooks call it (allocation + enqueue + the single
    #      host sync + slicing), median of 20 ---------------------------------------------------------------
    if extra:
        lat, ret = [], []
        for _ in range(25):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = vc.vidcom2_compression(x, model="qwen2_5_vl", base_scale=base, frame_token_len=N)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e6)
            ret.append((t1 - t0) * 1e6)
            del r
        out["one_shot_call_us"] = round(statistics.median(lat[5:]), 1)       # until the kept rows are written (device sync)
        # until the call RETURNS: the count comes from the selection launch's pinned-host mirror, so the caller goes on
        # (slicing, its next launches: stream-ordered) while the gather launch is still running
        out["one_shot_return_us"] = round(statistics.median(ret[5:]), 1)
    # ---- side: the same pass with plain correctly-rounded reductions ("exact" mode) ---------------------
    if extra and dtype != torch.float32:
        _ffi.set_mode("exact")
        for _ in range(args.warmup):
            step()
        e3 = time_steps(step, args.steps, False)
        _ffi.set_mode("torch")
        out["exact_mode"] = {"ms_per_step": round(e3 / args.steps * 1e3, 4),
                             "tokens_per_s": round(F * N / (e3 / args.steps), 1),
                             "pass_alg_GBs": round(alg_bytes_pass(F, N, D, es, base) / (e3 / args.steps) / 1e9, 1)}
    # ---- side: the opt-in fast variant (empirical margins only: no parity claim) and the proven-margin variant of the
    #      default mode ---------------------------------------------------------------------
    if extra and dtype != torch.float32:
        for mname, key in (("torch_fast", "fast_mode_no_parity_claim"), ("torch_proven", "proven_mode")):
            _ffi.set_mode(mname)
            for _ in range(args.warmup):
                step()
            e4 = time_steps(step, args.steps, False)
            _ffi.set_mode("torch")
            out[key] = {"ms_per_step": round(e4 / args.steps * 1e3, 4), "tokens_per_s": round(F * N / (e4 / args.steps), 1)}
    # ---- side: cfg2 (LLaVA-OV shape) on one GPU ------------------------------------------------------
    if extra and args.workload == "target":
        F2, N2, D2, dt2, b2 = WORKLOADS["cfg2"]
        x2 = x[: F2 * N2]                       # first 32 frames of the same tensor
        p2 = vc.vidcom2.CompressPlan(F2, N2, D2, dt2, dev, b2)
        for _ in range(args.warmup):
            p2.enqueue(x2)
        e2 = time_steps(lambda: p2.enqueue(x2), args.steps, False)
        out["cfg2"] = {"workload": "32x196x3584 bf16 retain 0.25", "ms_per_step": round(e2 / args.steps * 1e3, 4),
                       "median_ms_per_step": round(median_step_ms(lambda: p2.enqueue(x2), max(args.steps, 20)), 4),
                       "tokens_per_s": round(F2 * N2 / (e2 / args.steps), 1),
                       "pass_alg_GBs": round(alg_bytes_pass(F2, N2, D2, 2, b2) / (e2 / args.steps) / 1e9, 1)}
    # ---- side: cfg3 (Qwen2.5-VL shape) ---------------------------------------------------------------
    if extra and args.workload == "target":
        F3, N3, D3, dt3, b3 = WORKLOADS["cfg3"]
        x3 = synth.make(F3, N3, D3, dt3, seed=0, dist="drift").to(dev)
        p3 = vc.vidcom2.CompressPlan(F3, N3, D3, dt3, dev, b3)
        for _ in range(args.warmup):
            p3.enqueue(x3)
        e3 = time_steps(lambda: p3.enqueue(x3), args.steps, False)
        out["cfg3"] = {"workload": "64x324x3584 bf16 retain 0.125", "ms_per_step": round(e3 / args.steps * 1e3, 4),
                       "tokens_per_s": round(F3 * N3 / (e3 / args.steps), 1),
                       "pass_alg_GBs": round(alg_bytes_pass(F3, N3, D3, 2, b3) / (e3 / args.steps) / 1e9, 1)}
        del x3, p3
    # ---- side: a working set beyond the 256 MiB Infinity Cache: 512 frames x 196 x 3584 bf16 = 360 MB of X (the
    #      long-video config on ONE GPU; the `drift` generator's data -- until round 4 this leg used N(0,1) noise, whose
    #      zero-mean channels put 8 % of the frame means on the replay list; a timing leg, parity is tested elsewhere)
    if extra and args.workload == "target":
        Fb = 512
        xb = x_long               # four 128-frame `drift` clips in a row (three different ones: a video of four scenes)
        pb = vc.vidcom2.CompressPlan(Fb, N, D, dtype, dev, base)
        pb.enqueue(xb)
        Kb = pb.finish().K
        for _ in range(5):
            pb.enqueue(xb)
        nb = max(10, args.steps // 2)
        eb = time_steps(lambda: pb.enqueue(xb), nb, False)
        kb = kernel_profile(lambda: pb.enqueue(xb), nb, _ffi)
        out["roofline_big"] = roofline_of(kb, Fb, N, D, es, Kb, "long512", 1)
        out["long512"] = {"workload": f"512x{N}x{D} {DT_NAME[dtype]} retain {base}, one GPU", "ms_per_step": round(eb / nb * 1e3, 4),
                          "tokens_per_s": round(Fb * N / (eb / nb), 1),
                          "pass_alg_GBs": round(alg_bytes_pass(Fb, N, D, es, base) / (eb / nb) / 1e9, 1),
                          "pass_frac_of_8TBs": round(alg_bytes_pass(Fb, N, D, es, base) / (eb / nb) / 1e9 / HBM_PEAK_GBS, 4),
                          "kernels_us": kb}
        del xb, pb, x_long
    # ---- side (f3): LLaVA's get_2dPool fused with sweep 1.  Projector output 128 x (27x27) x 3584 bf16, bilinear pool
    #      to 14x14 = 196 tokens: torch's permute/interpolate/permute on the device + the full pass, against
    #      fused.pool_stats + the pass without its first sweep -------------------------------------------------
    if extra and args.workload == "target" and dtype != torch.float32:
        from vidcom2_amd.fused import pool_stats
        Hs = 27                                      # LLaVA-OneVision: 27 x 27 patches -> bilinear -> 14 x 14 = 196
        xin = torch.randn(F, Hs * Hs, D, device=dev, dtype=torch.float32).to(dtype)
        pl_a = vc.vidcom2.CompressPlan(F, N, D, dtype, dev, base)

        def unfused():
            t = xin.view(F, Hs, Hs, D).permute(0, 3, 1, 2).contiguous()
            t = torch.nn.functional.interpolate(t, size=[14, 14], mode="bilinear").permute(0, 2, 3, 1).reshape(F * N, D)
            pl_a.enqueue(t)

        def fused():
            pooled, ws = pool_stats(xin, Hs, Hs, "bilinear")
            pl_b = vc.vidcom2.CompressPlan(F, N, D, dtype, dev, base, ws=ws)
            pl_b.enqueue(pooled.view(F * N, D), have_stats=True)

        def pool_only():
            pool_stats(xin, Hs, Hs, "bilinear")
        res = {}
        for name, fn in (("torch_pool_then_pass", unfused), ("fused_pool_stats_then_pass", fused),
                         ("pool_stats_kernel_only", pool_only)):
            for _ in range(5):
                fn()
            res[name + "_us"] = round(time_steps(fn, max(10, args.steps // 2), False) / max(10, args.steps // 2) * 1e6, 1)
        res["workload"] = f"{F}x({Hs}x{Hs})x{D} {DT_NAME[dtype]} -> bilinear pool (get_2dPool) -> {F}x{N}x{D}, retain {base}"
        res["pool_alg_bytes"] = F * (Hs * Hs + N) * D * es
        res["pool_alg_GBs"] = round(res["pool_alg_bytes"] / (res["pool_stats_kernel_only_us"] * 1e-6) / 1e9, 1)
        out["llava_pool_fused"] = res
        del xin, pl_a
    # ---- side: the fused hook operations at a Qwen2.5-VL-7B prefill shape: 64 frames x 324 video tokens + 96 text
    #      tokens, 3584-d bf16, 12.5 % of the video kept -- the keep-list construction (vc2_keep_positions, one
    #      workgroup) and the ONE launch that writes text rows + kept video rows (vc2_gather_scatter) ------------
    if extra and args.workload == "target":
        from vidcom2_amd.fused import gather_scatter, keep_positions
        nvid, ntext = 64 * 324, 96
        Sq = nvid + ntext
        vm = torch.zeros(Sq, dtype=torch.bool, device=dev)
        vm[32:32 + nvid] = True
        kept = torch.arange(0, nvid, 8, device=dev, dtype=torch.int64)
        emb = torch.randn(Sq, D, device=dev, dtype=torch.float32).to(dtype)
        keep, _ = keep_positions(vm, kept, nvid)
        st_word = torch.zeros(1, dtype=torch.int32, device=dev)
        t_keep = time_steps(lambda: keep_positions(vm, kept, nvid), 20, False) / 20
        outb = gather_scatter([emb], keep, status=st_word)
        t_gs = time_steps(lambda: gather_scatter([emb], keep, dsts=outb, status=st_word), 50, False) / 50
        nbytes = 2 * keep.numel() * D * es + 8 * keep.numel()
        out["hook_fused_ops"] = {"shape": f"{Sq} positions ({nvid} video, {kept.numel()} kept), {D}-d {DT_NAME[dtype]}",
                                 "keep_positions_us_incl_counts": round(t_keep * 1e6, 1),
                                 "gather_scatter_us": round(t_gs * 1e6, 1), "gather_scatter_bytes": nbytes,
                                 "gather_scatter_GBs": round(nbytes / t_gs / 1e9, 1)}
        del emb, outb
    # ---- side: two clips in flight, one stream each (serving / batched eval) --------------------------
    if extra and args.workload == "target":
        streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
        plans = [plan, vc.vidcom2.CompressPlan(F, N, D, dtype, dev, base)]
        # a DIFFERENT clip on the second stream: with the same tensor on both, X stays in the 256 MiB Infinity Cache
        # and the leg reads 13 % better than two real clips do
        xs2 = [x, synth.make(F, N, D, dtype, seed=1, dist="drift").to(dev)]

        def two():
            for st, pl, xi in zip(streams, plans, xs2):
                with torch.cuda.stream(st):
                    pl.enqueue(xi)
        for _ in range(args.warmup):
            two()
        e4 = time_steps(two, args.steps, False)
        out["two_clips_in_flight"] = {"ms_per_round": round(e4 / args.steps * 1e3, 4),
                                      "tokens_per_s": round(2 * F * N / (e4 / args.steps), 1)}
        # ... and THREE: the throughput operating point of a serving loop -- one clip's 60 us chain of
        # single-workgroup kernels runs under the sweeps of the two others.  Three DIFFERENT clips (513 MiB: X no longer
        # fits the Infinity Cache, as with real clips); per-clip time, tokens/s and fraction of the 8 TB/s roofline
        streams4 = streams + [torch.cuda.Stream(), torch.cuda.Stream()]
        plans4 = plans + [vc.vidcom2.CompressPlan(F, N, D, dtype, dev, base) for _ in range(2)]
        xs4 = xs2 + [synth.make(F, N, D, dtype, seed=sd, dist="drift").to(dev) for sd in (2, 3)]
        for lanes, key in ((3, "three_clips_in_flight"), (4, "four_clips_in_flight")):
            def many():
                for st, pl, xi in zip(streams4[:lanes], plans4[:lanes], xs4[:lanes]):
                    with torch.cuda.stream(st):
                        pl.enqueue(xi)
            for _ in range(args.warmup):
                many()
            e5 = min(time_steps(many, args.steps, False) for _ in range(2))
            per_clip = e5 / args.steps / lanes
            out[key] = {"us_per_clip": round(per_clip * 1e6, 2), "tokens_per_s": round(F * N / per_clip, 1),
                        "pass_alg_GBs": round(alg_bytes_pass(F, N, D, es, base) / per_clip / 1e9, 1),
                        "pass_frac_of_8TBs": round(alg_bytes_pass(F, N, D, es, base) / per_clip / 1e9 / HBM_PEAK_GBS, 4),
                        "note": f"{lanes} different clips, one stream and one plan each; inputs resident"}
        del plans4, xs4

    # ---- side: the other dtypes of BASELINE.json's shapes, each behind its own parity gate (C++ oracle, same tensor):
    #      the target shape in fp32, and ONE clip of the batched-eval config (128 x 196 x 4096 fp16) --------------
    if extra and args.workload == "target":
        import oracle
        oracle.set_mode("torch")
        for wl in ("target_fp32", "target_f16", "cfg5clip"):
            Fo, No, Do, dto, bo = WORKLOADS[wl]
            xo_cpu = synth.make(Fo, No, Do, dto, seed=0, dist="drift")
            xo = xo_cpu.to(dev)
            po = vc.vidcom2.CompressPlan(Fo, No, Do, dto, dev, bo)
            po.enqueue(xo)
            ro = po.finish()
            leg = {"workload": f"{Fo}x{No}x{Do} {DT_NAME[dto]} retain {bo}"}
            if not args.no_cpu_baseline:
                ref = oracle.compress_indices(xo_cpu, No, bo)
                leg["parity"] = bool(ro.ks.cpu().tolist() == ref["ks"].tolist() and torch.equal(ro.global_idx.cpu(), ref["global_idx"]))
                if not leg["parity"]:
                    raise SystemExit(f"[bench] PARITY FAILURE ({wl}): kept indices / budgets differ from the oracle")
            # (the oracle kept the host busy and the GPU idle for seconds: warm up until the clocks are back -- without
            # this the leg measured up to 7x slow)
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.3:
                for _ in range(max(args.warmup, 10)):
                    po.enqueue(xo)
                torch.cuda.synchronize()
            eo = min(time_steps(lambda: po.enqueue(xo), args.steps, False) for _ in range(2))
            eso = 4 if dto == torch.float32 else 2
            leg.update({"ms_per_step": round(eo / args.steps * 1e3, 4), "tokens_per_s": round(Fo * No / (eo / args.steps), 1),
                        "pass_alg_GBs": round(alg_bytes_pass(Fo, No, Do, eso, bo) / (eo / args.steps) / 1e9, 1),
                        "pass_frac_of_8TBs": round(alg_bytes_pass(Fo, No, Do, eso, bo) / (eo / args.steps) / 1e9 / HBM_PEAK_GBS, 4)})
            out[wl] = leg
            if wl == "cfg5clip":
                # BASELINE config 5 on ONE GPU: the whole batch of 16 clips through compress_batch (its default lanes: a clip's
                # single-workgroup kernels overlap the other clip's sweeps).  Four distinct clips, four times each.
                clips4 = [xo] + [synth.make(Fo, No, Do, dto, seed=sd, dist="drift").to(dev) for sd in (1, 2, 3)]
                batch = [clips4[i % 4] for i in range(CFG5_CLIPS)]
                rb = vc.vidcom2.compress_batch(batch, No, bo)
                if not (rb[0].K == ro.K and torch.equal(rb[0].global_idx, ro.global_idx) and torch.equal(rb[4].global_idx, ro.global_idx)):
                    raise SystemExit("[bench] PARITY FAILURE (cfg5_batch16): the batch's clip 0 differs from the single pass")
                del rb
                for _ in range(3):
                    vc.vidcom2.compress_batch(batch, No, bo)
                nbat = max(5, args.steps // 4)
                ebat = min(time_steps(lambda: vc.vidcom2.compress_batch(batch, No, bo), nbat, False) for _ in range(2)) / nbat
                out["cfg5_batch16"] = {"workload": f"{CFG5_CLIPS} clips x {Fo}x{No}x{Do} {DT_NAME[dto]} retain {bo}, compress_batch(in_flight={vc.vidcom2.BATCH_LANES}), one GPU",
                                       "ms_per_batch": round(ebat * 1e3, 3), "us_per_clip": round(ebat / CFG5_CLIPS * 1e6, 1),
                                       "tokens_per_s": round(CFG5_CLIPS * Fo * No / ebat, 1),
                                       "pass_frac_of_8TBs": round(CFG5_CLIPS * alg_bytes_pass(Fo, No, Do, 2, bo) / ebat / 1e9 / HBM_PEAK_GBS, 4),
                                       "vs_one_clip_at_a_time": round(leg["ms_per_step"] * 1e-3 * CFG5_CLIPS / ebat, 3)}
                del clips4, batch
            del xo, po, xo_cpu
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
