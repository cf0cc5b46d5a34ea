"""Stage-by-stage comparison of the HIP path with the CPU oracle (development aid; the pytest
suite in tests/ is the real gate).  Run on a GPU box:  python tests/tools/gpu_check.py [--big]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle as O
import vidcom2_amd as vc
from vidcom2_amd import synth, _ffi
from vidcom2_amd._ffi import lib, ptr, stream_ptr, DTYPE_CODE, check

dev = torch.device("cuda:0")


def cnt(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return int(((a != b) & ~(a.isnan() & b.isnan())).sum())


def run_case(F, N, D, dt, seed, dist, base):
    x = synth.make(F, N, D, dt, seed, dist)
    xd = x.to(dev)
    ref = O.compress_indices(x, N, base)
    _, rvar = O.select_low_var_channel_idx(x)
    # stage: variance
    var_T, var_f = vc.vidcom2._channel_variance(xd)
    res = {"var": cnt(var_T, rvar)}
    # stage: channel order (API)
    order = vc.vidcom2.low_var_channel_order(xd)
    res["cidx"] = int((order.cpu() != ref["chan_idx"]).sum())
    # stage: device mask vs oracle set
    mask = torch.zeros(D, dtype=torch.uint8, device=dev)
    check(lib().vc2_chan_select(ptr(var_f), D, int(D * 0.5), ptr(mask), None, None, None, None, None, stream_ptr(dev)), "chan_select")
    rmask = torch.zeros(D, dtype=torch.uint8); rmask[ref["chan_idx"]] = 1
    res["mask"] = int((mask.cpu() != rmask).sum())
    # full pass
    got = vc.compress(xd, N, base, want_scores=True)
    res["v"] = cnt(got.v_score, ref["v"]); res["f"] = cnt(got.f_score, ref["f"])
    res["ks"] = int(got.ks.cpu().tolist() != ref["ks"].tolist())
    gi = got.global_idx.cpu()
    res["idx"] = -1 if gi.numel() != ref["global_idx"].numel() else int((gi != ref["global_idx"]).sum())
    res["rows"] = int(not torch.equal(got.rows.cpu(), x[gi]))
    mx = max((got.v_score.cpu().double() - ref["v"].double()).abs().max().item(),
             (got.f_score.cpu().double() - ref["f"].double()).abs().max().item())
    ok = res["ks"] == 0 and res["idx"] == 0 and res["mask"] == 0 and res["rows"] == 0 and res["cidx"] == 0
    print(str(dt)[6:], (F, N, D, base), seed, dist, res, "maxerr %.2e" % mx, "K", got.K, "OK" if ok else "DIFF", flush=True)
    return ok


def main():
    big = "--big" in sys.argv
    mode = "exact" if "--exact" in sys.argv else "torch"
    O.set_mode(mode); _ffi.set_mode(mode)
    print("mode:", mode, flush=True)
    cases = []
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        for shp in [(4, 49, 64, .25), (8, 196, 1024, .25), (3, 50, 72, .3), (16, 169, 3584, .15), (32, 196, 3584, .25)]:
            for seed in (0, 1):
                for dist in ("drift", "iid"):
                    cases.append(shp[:3] + (dt, seed, dist, shp[3]))
    if big:
        cases += [(64, 324, 3584, torch.bfloat16, 0, "drift", .125), (128, 196, 3584, torch.bfloat16, 0, "drift", .25),
                  (128, 196, 3584, torch.float32, 0, "drift", .25), (16, 196, 4096, torch.float16, 0, "drift", .25)]
    n_ok = 0
    for c in cases:
        try:
            n_ok += bool(run_case(*c))
        except Exception as e:  # noqa
            print("EXC", c, repr(e), flush=True)
    print(f"{n_ok}/{len(cases)} cases OK")


if __name__ == "__main__":
    main()
