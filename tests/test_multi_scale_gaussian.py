"""Standalone `_multi_scale_gaussian` (vidcom2.py:59-62): oracle and HIP path against the reference's
recorded outputs (tests/golden/msg_cases.json, made by make_msg_golden.py)."""
import pytest
import torch

import _stub_models as S
import oracle as O
from conftest import load_json
from vidcom2_amd import synth

CASES = load_json("msg_cases.json")
TOL_F32 = 1e-5          # north_star tolerance for fp32 scores (torch's vectorised expf is not correctly rounded)


def _cid(c):
    return f"{c['dt']}-{c['F']}x{c['N']}x{c['C']}-{c['kind']}-a{len(c['alphas'])}"


def _check(c, got):
    got = got.cpu()
    assert got.shape == (c["F"], c["N"]) and got.dtype == S.DT[c["dt"]]
    if c["dt"] == "f32":
        flat = got.reshape(-1).double()
        want = torch.tensor(c["sample"], dtype=torch.float64)
        assert (flat[torch.tensor(c["sample_idx"])] - want).abs().max().item() <= TOL_F32
    else:
        assert synth.sha256_tensor(got) == c["sha"]          # bit-exact in half precision


@pytest.mark.parametrize("mode", ["torch", "exact"])
@pytest.mark.parametrize("c", CASES, ids=_cid)
def test_oracle_matches_reference(c, mode):
    O.set_mode(mode)
    try:
        x, cen = S.msg_inputs(c)
        _check(c, O.multi_scale_gaussian(x, cen, c["alphas"]))
    finally:
        O.set_mode("torch")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["torch", "exact"])
@pytest.mark.parametrize("c", CASES, ids=_cid)
def test_hip_matches_reference_and_oracle(c, mode):
    from vidcom2_amd import _ffi
    from vidcom2_amd.vidcom2 import _multi_scale_gaussian
    old = _ffi.get_mode()
    _ffi.set_mode(mode)
    O.set_mode(mode)
    try:
        x, cen = S.msg_inputs(c)
        got = _multi_scale_gaussian(x.cuda(), cen.cuda(), c["alphas"])
        _check(c, got)
        assert torch.equal(got.cpu(), O.multi_scale_gaussian(x, cen, c["alphas"]))     # all dtypes, bit-exact
    finally:
        _ffi.set_mode(old)
        O.set_mode("torch")


@pytest.mark.gpu
def test_hip_composes_like_the_reference():
    """compute_gaussian_scores == normalize -> centres -> two helper calls (vidcom2.py:45-57), with the
    helper fed the device pass's own x^ and centres rebuilt from its outputs."""
    import vidcom2_amd as V
    from vidcom2_amd.vidcom2 import _multi_scale_gaussian
    x = synth.make(6, 36, 128, torch.bfloat16, 9).cuda()
    sel = V.select_low_var_channels(x)
    v, f = V.compute_gaussian_scores(sel, 36)
    d = O.gaussian_debug(sel.cpu(), torch.arange(sel.shape[1]), 36)
    xh = (sel.cpu().float() / d["norm"].float()[:, None]).to(torch.bfloat16).reshape(6, 36, -1)
    alphas = [2 ** k for k in range(-3, 2)]
    v2 = _multi_scale_gaussian(xh.cuda(), d["vid_center"].reshape(1, 1, -1).cuda(), alphas)
    f2 = _multi_scale_gaussian(xh.cuda(), d["frame_center"].reshape(6, 1, -1).cuda(), alphas)
    assert torch.equal(v2, v) and torch.equal(f2, f)


def test_argument_errors():
    from vidcom2_amd.vidcom2 import _multi_scale_gaussian
    with pytest.raises(RuntimeError):          # not broadcastable: torch's own error for x - center
        _multi_scale_gaussian(torch.zeros(2, 3, 4), torch.zeros(3, 1, 4), [1.0])
    assert _multi_scale_gaussian(torch.zeros(2, 3, 4), torch.zeros(1, 1, 4), []) == 0
    with pytest.raises(RuntimeError, match="no CPU fallback|No CPU|CPU fallback"):
        _multi_scale_gaussian(torch.zeros(2, 3, 4), torch.zeros(1, 1, 4), [1.0])
