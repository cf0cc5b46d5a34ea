"""Bounded soak (promoted from tests/tools/soak_gpu_vs_oracle.py): the HIP pass (flag-and-replay) against the oracle
(which replays torch's accumulation order for EVERY token) on 512 seeds no fixture has seen, at the target's
channel width, bf16 and fp16, `torch` mode: kept indices, budgets and both score tensors bit-exact."""
import pytest
import torch

import oracle as O
from vidcom2_amd import synth

pytestmark = pytest.mark.gpu
SHAPES = [(16, 196, 3584), (8, 324, 3584), (12, 169, 3584), (6, 196, 4096)]
SEEDS = range(5000, 5032)


@pytest.mark.parametrize("dist", ["iid", "drift"])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_fresh_seeds_match_the_oracle(shape, dt, dist):
    from vidcom2_amd import _ffi
    from vidcom2_amd.vidcom2 import compress
    F, N, D = shape
    assert _ffi.get_mode() == "torch"
    O.set_mode("torch")
    bad = []
    for seed in SEEDS:
        x = synth.make(F, N, D, dt, seed, dist)
        r = compress(x.cuda(), N, 0.25, want_scores=True)
        o = O.compress_indices(x, N, 0.25)
        if not (torch.equal(r.global_idx.cpu(), o["global_idx"]) and torch.equal(r.ks.cpu(), o["ks"])
                and torch.equal(r.v_score.cpu(), o["v"]) and torch.equal(r.f_score.cpu(), o["f"])):
            bad.append(seed)
    assert not bad, f"seeds {bad} differ from the oracle at {shape} {dt} {dist}"
    counters = _ffi.lib().vc2_selftest_counters
    import ctypes
    out = (ctypes.c_int32 * 8)()
    assert counters(out, 0) == 0 and not any(out), list(out)            # no bounded loop of the engine expired
