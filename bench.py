#!/usr/bin/env python
"""bench.py -- throughput of the VidCom2 token-compression pass on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" = one whole pass over one batch of synthetic frame-token embeddings already resident in
HBM: channel selection, scoring, budgets, per-frame selection and the kept-row gather
(vidcom2.py:15-36).  Metric (BASELINE.json): input video tokens compressed per second at the
stated retain ratio, with kept indices / budgets bit-exact vs the reference oracle (checked here on
every run before timing).

N = 1 : workload "target" = 128 frames x 196 tokens x 3584-d bf16, 25 % retain (the shape
        BASELINE.json's north-star target is quoted on); cfg2 (32x196x3584) is reported beside it.
N > 1 : weak scaling -- every rank holds 128 frames of ONE long video of 128*N frames, frame-sharded
        with three small RCCL all-gathers (channel stats, centre sums, per-frame uniqueness scores).

One JSON line on rank 0 (see the repo prompt for the contract), with two extra objects:
  "roofline"     the dominant kernel's achieved algorithmic HBM rate (hipEvent-timed inside this run)
  "cpu_baseline" the CPU oracle ("port") timed on this box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s

WORKLOADS = {
    # name: (F, N, D, dtype, base_scale)
    "target": (128, 196, 3584, torch.bfloat16, 0.25),
    "cfg2": (32, 196, 3584, torch.bfloat16, 0.25),
    "cfg3": (64, 324, 3584, torch.bfloat16, 0.125),
    "target_fp32": (128, 196, 3584, torch.float32, 0.25),
    "cfg5clip": (128, 196, 4096, torch.float16, 0.25),
    "cfg1": (8, 196, 1024, torch.float32, 0.25),
}
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
        "k_dist": X + 12 * F * N,                # sweep 3: read X + den, write two distances
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


def cpu_baseline(x_cpu, N, base, budget_s=12.0):
    """The CPU oracle (kind 'port': C++/OpenMP restatement of vidcom2.py, proven equal to the imported
    reference on the golden fixtures) timed on this box's host cores on the same workload."""
    import oracle
    cores = os.cpu_count() or 1
    oracle.set_num_threads(cores)
    oracle.set_mode("torch")             # same semantics as the HIP path's default mode
    oracle.compress_indices(x_cpu[: 8 * N], N, base)   # warm (library load, page-in)
    t0 = time.perf_counter()
    reps = 0
    while True:
        o = oracle.compress_indices(x_cpu, N, base)
        _ = x_cpu[o["global_idx"]]
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps
    return {"value": x_cpu.shape[0] / dt, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{reps} full passes of the same workload ({x_cpu.shape[0]} tokens each), {dt * 1e3:.1f} ms/pass"}, o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg2 side measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("VC2_BENCH_FORCE_DIST") == "1"   # (the override exercises the
    #                                 sharded code path on one GPU; used by the single-GPU smoke run only)
    if args.gpus != world and rank == 0 and dist_on:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # lazy communicator creation on purpose: with `device_id=` (eager init) every pass that uses the library's
        # side stream measured ~45 us slower on this stack (323 vs 278 us), with it the numbers match a plain process
        torch.distributed.init_process_group("nccl")

    import vidcom2_amd as vc
    from vidcom2_amd import _ffi, synth

    F, N, D, dtype, base = WORKLOADS[args.workload]
    es = 4 if dtype == torch.float32 else 2
    F_total = F * world

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

    if not dist_on:
        plan = vc.vidcom2.CompressPlan(F, N, D, dtype, dev, base)
        step = lambda: plan.enqueue(x)          # noqa: E731
        finish = plan.finish
    else:
        from vidcom2_amd.sharded import ShardedCompressor
        sc = ShardedCompressor(F, N, D, dtype, dev, base, group=None)
        step = lambda: sc.enqueue(x)            # noqa: E731
        finish = sc.finish

    # ---- parity gate: kept indices + budgets must equal the oracle's before anything is timed ------
    cpu = None
    step()
    res = finish()
    if not dist_on and not args.no_cpu_baseline:
        cpu, ref = cpu_baseline(x_cpu, N, base)
        ok = res.ks.cpu().tolist() == ref["ks"].tolist() and torch.equal(res.global_idx.cpu(), ref["global_idx"])
        if not ok:
            raise SystemExit("[bench] PARITY FAILURE: kept indices / budgets differ from the oracle")
    K = res.K

    # ---- timed region -------------------------------------------------------------------------------
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
    roof = None
    kern = {}
    _ffi.profile_enable(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof = _ffi.profile_collect()
    _ffi.profile_enable(False)
    for name, (tot, cnt) in prof.items():
        kern[name] = round(tot / cnt * 1e3, 2)          # us per launch
    sweeps = {n: kern[n] for n in ("k_chan_stats", "k_norm_colsum", "k_dist", "k_gather_rows") if n in kern}
    if sweeps:
        dom = max(sweeps, key=sweeps.get)
        ab = kernel_alg_bytes(dom, F, N, D, es, K)
        ach = ab / (sweeps[dom] * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, args.workload, world),
                "alg_bytes_per_launch": ab, "avg_us": sweeps[dom]}

    out = {
        "metric": "video-tokens compressed/sec at 25% retain; kept-index bit-exact vs ref",
        "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DT_NAME[dtype], "data": "synthetic",
        "config": {"workload": f"{args.workload}: {F_total} frames x {N} tokens x {D}-d {DT_NAME[dtype]}, "
                               f"retain {base}" + (f", frame-sharded {F} frames/GPU, 3 RCCL all-gathers" if dist_on else ""),
                   "kept_tokens": K, "parallelism": f"frame-shard x{world}" if dist_on else "single GPU"},
        "roofline": roof,
        "cpu_baseline": cpu,
        "pass_roofline": {"alg_bytes": alg_bytes_pass(F, N, D, es, base) * world,
                          "achieved_GBs": round(alg_bytes_pass(F, N, D, es, base) * world / (ms * 1e-3) / 1e9, 1),
                          "frac_of_8TBs_per_gpu": round(alg_bytes_pass(F, N, D, es, base) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "kernels_us": kern,
    }

    out["mode"] = _ffi.get_mode() + (" (bit-exact to the CPU reference: boundary-fragile tokens replay torch's fp32 "
                                     "accumulation order)" if _ffi.get_mode() == "torch" else "")
    # ---- side measurement: the same pass with plain correctly-rounded reductions ("exact" mode) ---------
    if not dist_on and not args.no_extra and dtype != torch.float32:
        _ffi.set_mode("exact")
        for _ in range(args.warmup):
            step()
        e3 = time_steps(step, args.steps, False)
        _ffi.set_mode("torch")
        out["exact_mode"] = {"ms_per_step": round(e3 / args.steps * 1e3, 4),
                             "tokens_per_s": round(F * N / (e3 / args.steps), 1),
                             "pass_alg_GBs": round(alg_bytes_pass(F, N, D, es, base) / (e3 / args.steps) / 1e9, 1)}
    # ---- side measurement: cfg2 (LLaVA-OV shape) on one GPU -----------------------------------------
    if not dist_on and not args.no_extra and args.workload == "target":
        F2, N2, D2, dt2, b2 = WORKLOADS["cfg2"]
        x2 = x[: F2 * N2]                       # first 32 frames of the same tensor
        p2 = vc.vidcom2.CompressPlan(F2, N2, D2, dt2, dev, b2)
        for _ in range(args.warmup):
            p2.enqueue(x2)
        e2 = time_steps(lambda: p2.enqueue(x2), args.steps, False)
        out["cfg2"] = {"workload": "32x196x3584 bf16 retain 0.25", "ms_per_step": round(e2 / args.steps * 1e3, 4),
                       "tokens_per_s": round(F2 * N2 / (e2 / args.steps), 1),
                       "pass_alg_GBs": round(alg_bytes_pass(F2, N2, D2, 2, b2) / (e2 / args.steps) / 1e9, 1)}

    # ---- side measurement: two clips in flight, one stream each (serving / batched eval) --------------
    # The single-workgroup selection replays leave the GPU mostly idle; a second clip on its own stream fills it.
    if not dist_on and not args.no_extra and args.workload == "target":
        # (the current stream plus ONE new one: each brings an internal side stream, and four streams is what the
        # default four hardware queues run without multiplexing)
        streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
        plans = [plan, vc.vidcom2.CompressPlan(F, N, D, dtype, dev, base)]

        def two():
            for st, pl in zip(streams, plans):
                with torch.cuda.stream(st):
                    pl.enqueue(x)
        for _ in range(args.warmup):
            two()
        e4 = time_steps(two, args.steps, False)
        out["two_clips_in_flight"] = {"ms_per_round": round(e4 / args.steps * 1e3, 4),
                                      "tokens_per_s": round(2 * F * N / (e4 / args.steps), 1)}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
