"""GPU: clips with more than 2^19 tokens (tests/golden/long_cases.json, made by tests/golden/make_long_golden.py from
the reference itself).  Beyond 2^19 rows torch's outer-sum cascade adds the video-centre mean with level_power 5
(blocks of 32 rows, SumKernel.cpp); the `cancel` fixtures are built so that the reference's centre IS that cascade's
value and differs from the 16-row one, i.e. a replay that kept blocks of 16 fails them.  Compared bit for bit:
budgets, kept indices, both score tensors -- unsharded in the default, the proven-margin and the replay-everything
mode, and frame-sharded as 8 / 4 / 2 logical ranks."""
import functools
import warnings

import pytest
import torch

import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth
from conftest import DT, load_json

pytestmark = pytest.mark.gpu
LONG = load_json("long_cases.json")["cases"]


def _id(c):
    return f"{c['name']}-{c['F']}x{c['N']}x{c['D']}-{c['dtype']}-{c['dist']}-s{c['seed']}"


@functools.lru_cache(maxsize=1)
def _input(F, N, D, dn, seed, dist):
    return synth.make(F, N, D, DT[dn], seed, dist)


def _x(c):
    x = _input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"])
    assert synth.sha256_tensor(x) == c["x_sha256"]
    return x


def _check(c, ks, gidx, v=None, f=None):
    assert synth.sha256_tensor(ks.to(torch.int64).cpu()) == c["ks_sha256"], "budgets differ from the reference"
    assert ks[:16].tolist() == c["ks_head"]
    assert gidx.numel() == c["K"]
    assert gidx[:8].tolist() == c["idx_head"] and gidx[-8:].tolist() == c["idx_tail"]
    assert synth.sha256_tensor(gidx) == c["idx_sha256"], "kept indices differ from the reference"
    if v is not None:
        assert synth.sha256_tensor(v) == c["v_sha256"] and synth.sha256_tensor(f) == c["f_sha256"], \
            "scores differ from the reference"


@pytest.mark.parametrize("mode", ["torch", "torch_proven"])
@pytest.mark.parametrize("c", LONG, ids=_id)
def test_long_clip_matches_the_reference(c, mode):
    """Budgets, kept indices and both score tensors equal the reference's -- in the default mode (every case, the fp16
    `cancel` clips included: their frame means are the ones the pre-round-4 default missed) and in the proven-margin mode."""
    x = _x(c)
    try:
        _ffi.set_mode(mode)
        with warnings.catch_warnings():
            warnings.simplefilter("error")          # the "beyond the modelled range" warning is gone for these sizes
            got = vc.compress(x.cuda(), c["N"], c["base"], want_scores=True)
    finally:
        _ffi.set_mode("torch")
    _check(c, got.ks.cpu(), got.global_idx.cpu(), got.v_score.cpu(), got.f_score.cpu())


@pytest.mark.parametrize("c", [c for c in LONG if c["name"] in ("long", "long_tailcols") and c["dist"] == "cancel"], ids=_id)
def test_long_clip_replaying_every_value(c):
    """Debug mode 2: every norm, distance and centre mean is replayed in torch's order -- all 128 video-centre
    columns go through the level-power-5 cascade (and, for C = 100, row_sum's four interleaved chains)."""
    x = _x(c)
    try:
        assert _ffi.lib().vc2_set_mode(2) == 0
        got = vc.compress(x.cuda(), c["N"], c["base"], want_scores=True)
    finally:
        _ffi.set_mode("torch")
    _check(c, got.ks.cpu(), got.global_idx.cpu(), got.v_score.cpu(), got.f_score.cpu())


@pytest.mark.parametrize("c", [c for c in LONG if c["name"] == "long_sharded"], ids=_id)
def test_long_clip_frame_sharded(c):
    """3072 frames as 8 / 4 / 2 logical ranks through the stage entry points the RCCL path calls: rows per rank are a
    multiple of 32, so exchange 2b carries 32-row block sums and every rank finishes the level-power-5 cascade."""
    from test_sharded import _emulate_ranks_on_one_gpu
    _ffi.set_mode("torch")
    F, N, D = c["F"], c["N"], c["D"]
    x = _x(c).cuda()
    whole = vc.compress(x, N, c["base"], want_scores=True)
    assert synth.sha256_tensor(whole.v_score.cpu()) == c["v_sha256"]
    total = (whole.v_score + whole.f_score).float().flatten()       # the reference's scores, token by token
    for P in (8, 4, 2):
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            res, st = _emulate_ranks_on_one_gpu(x, F, N, D, DT[c["dtype"]], c["base"], P, torch.device("cuda:0"),
                                                 vc_cap=D // 2)
        _check(c, torch.cat([r.ks for r in res]).cpu(), torch.cat([r.global_idx for r in res]).cpu())
        assert all(s_.vc_fragile == 0 for s_ in st)          # every boundary-near video-centre mean was replayed
        assert torch.equal(torch.cat([s_.total for s_ in st]), total), f"P={P}"
        del res, st
    # 3 ranks of 1024 frames hold 200704 = 32 * 6272 rows each as well; 6 ranks of 512 frames likewise
    res, st = _emulate_ranks_on_one_gpu(x, F, N, D, DT[c["dtype"]], c["base"], 3, torch.device("cuda:0"), vc_cap=D // 2)
    _check(c, torch.cat([r.ks for r in res]).cpu(), torch.cat([r.global_idx for r in res]).cpu())
    assert torch.equal(torch.cat([s_.total for s_ in st]), total)
    _ffi.set_mode("torch")
