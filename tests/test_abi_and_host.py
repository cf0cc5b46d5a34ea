"""CPU: the C-ABI library loads and exports every symbol include/vc2.h declares; host-side logic of the
plugin API (argument validation, error behaviour, capacity bound, host top-k order)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import oracle as O
import vidcom2_amd as vc
from vidcom2_amd import _ffi, synth

from conftest import DT, ROOT, load_core_cases


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "vc2.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vc2_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _header_symbols()
    assert len(names) >= 20
    handle = ctypes.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/vc2.h but not exported by libvc2hip.so"
    # and the Python binding covers the same set
    assert sorted(_ffi.EXPORTED_SYMBOLS) == names
    assert b"gfx950" in _ffi.lib().vc2_version()


def test_reference_api_surface():
    # token_compressor/vidcom2/__init__.py:2-22 names + MODEL_SPECS (vidcom2.py:7-13)
    for n in ("vidcom2_compression", "select_low_var_channels", "compute_gaussian_scores", "compute_scales",
              "select_outlier_indices", "map_features", "_map_linear_offset", "_map_grid_vid"):
        assert n in vc.__all__ and callable(getattr(vc, n))
    assert vc.MODEL_SPECS["llava_ov"] == {"tpf": 196, "mapper": "linear"}
    assert vc.MODEL_SPECS["llava_vid"] == {"tpf": 169, "mapper": "grid_vid", "grid": 13}
    assert all(vc.MODEL_SPECS[m]["tpf"] is None for m in ("qwen2_vl", "qwen2_5_vl", "qwen3_vl"))
    from vidcom2_amd.vidcom2 import _multi_scale_gaussian  # noqa: F401  (importable like the reference's)


def test_error_behaviour_matches_reference():
    x = torch.zeros(20, 8)
    with pytest.raises(ValueError, match="Unknown model: nope"):
        vc.vidcom2_compression(x, model="nope")
    with pytest.raises(ValueError, match="frame_token_len required for qwen2_5_vl"):
        vc.vidcom2_compression(x, model="qwen2_5_vl")
    with pytest.raises(ValueError, match="img_feat required for grid mapping"):
        vc.vidcom2_compression(torch.zeros(338, 8), model="llava_vid")
    with pytest.raises(ValueError, match="img_feat required for grid mapping"):
        vc.map_features([torch.zeros(1, dtype=torch.long)], x, None, vc.MODEL_SPECS["llava_vid"])


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing on the host."""
    x = synth.make(2, 10, 8, torch.float32, 0, "iid")
    for fn in (lambda: vc.vidcom2_compression(x, model="qwen2_vl", frame_token_len=10),
               lambda: vc.select_low_var_channels(x), lambda: vc.compute_gaussian_scores(x, 10),
               lambda: vc.compute_scales(torch.zeros(4), 0.25),
               lambda: vc.select_outlier_indices(torch.zeros(2, 10), torch.zeros(2), 10)):
        with pytest.raises(RuntimeError, match="no CPU fallback|No CPU|CPU fallback"):
            fn()


def test_workspace_and_limits():
    assert _ffi.workspace_bytes(128, 196, 3584, torch.bfloat16) < 64 << 20
    with pytest.raises(NotImplementedError):
        _ffi.workspace_bytes(2, 10, 16384, torch.float32)       # > 1024 column vectors per row
    with pytest.raises(RuntimeError):
        _ffi.workspace_bytes(0, 10, 64, torch.float32)


def test_kept_capacity_bounds_every_fixture():
    L = _ffi.lib()
    for c in load_core_cases():
        cap = L.vc2_kept_capacity(c["F"], c["N"], c["base"])
        assert c["K"] <= cap <= c["F"] * c["N"]
    assert L.vc2_kept_capacity(128, 196, 0.25) < 0.3 * 128 * 196


def test_synth_is_deterministic():
    a = synth.make(3, 7, 16, torch.bfloat16, 5, "drift")
    b = synth.make(3, 7, 16, torch.bfloat16, 5, "drift")
    assert torch.equal(a, b) and not torch.equal(a, synth.make(3, 7, 16, torch.bfloat16, 6, "drift"))
    x32 = synth.make_fp32(4, 5, 8, 1)
    assert np.array_equal(x32[2:4], synth.make_fp32_frames(4, 5, 8, 2, 2, 1))
    bits = synth.f32_to_bf16_bits(np.array([1.0, 1.00390625, 1.01171875, np.inf, -0.0], np.float32))
    assert bits.tolist() == [0x3F80, 0x3F80, 0x3F82, 0x7F80, 0x8000]    # ties-to-even


def test_fused_ops_have_no_cpu_fallback():
    """f1 / f3 host mirrors: argument checks run anywhere, the work itself only on a ROCm device."""
    from vidcom2_amd import fused
    x = torch.zeros(4, 8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback|CPU fallback"):
        fused.gather_scatter([x])
    with pytest.raises(ValueError):
        fused.gather_scatter([])
    with pytest.raises(RuntimeError, match="no CPU fallback|CPU fallback"):
        fused.keep_positions(torch.zeros(5, dtype=torch.bool), torch.zeros(0, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU fallback|CPU fallback"):
        fused.pool_stats(torch.zeros(2, 16, 8, dtype=torch.bfloat16), 4, 4, "average")
    assert set(fused.POOL_MODES) == {"average", "max", "bilinear"}


def test_wait_host_count_is_a_plain_host_spin():
    """vc2_wait_host_count (the host side of vc2_compress_ex2's pinned mirror) touches no device: it returns the word once
    it is non-negative and a negative number when it stays at the caller's -1 for longer than the timeout."""
    import numpy as np
    w = np.array([-1, 0], dtype=np.int64)
    L = _ffi.lib()
    assert L.vc2_wait_host_count(ctypes.c_void_p(w.ctypes.data), 0.01) < 0
    w[0] = 6272
    assert L.vc2_wait_host_count(ctypes.c_void_p(w.ctypes.data), 0.01) == 6272
    assert L.vc2_wait_host_count(None, 0.01) < 0


def test_bench_issues_a_constant_number_of_passes_per_rank_before_timing():
    """bench.py's clock-warming passes: with more than one rank their number must not depend on the rank's own clock
    (every pass of the frame-sharded path is four all-gathers; round 4: a loop bounded by `perf_counter()` ran ten passes
    more on one rank than on the other now and then, and both hung)."""
    import importlib.util
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vc2_bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    counts = []
    for delay in (0.0, 0.02):                                   # a fast and a slow "rank"
        n = [0]

        def step():
            n[0] += 1
            time.sleep(delay / 10)
        issued = bench.warm_clocks(step, True, sync=lambda: None)
        assert issued == n[0]
        counts.append(issued)
    assert counts[0] == counts[1] == 50
    n = [0]
    assert bench.warm_clocks(lambda: n.__setitem__(0, n[0] + 1), False, sync=lambda: None, seconds=0.01) == n[0] >= 10


def test_f16_fast_quotient_is_the_ieee_quotient(tmp_path):
    """The fp16 fast path of sweep 2 (k_norm_colsum2) divides by q = fma(e, r, q0), q0 = x r, e = fma(-dn, q0, x): the
    exhaustive checker behind it (tests/tools/check_f16_quotient.c; every fp16 x, a strided subset of the denominators
    here, r off by up to +-4 ulps) must find the IEEE fp32 quotient every time."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = tmp_path / "chk"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", os.path.join(ROOT, "tests", "tools", "check_f16_quotient.c"),
                           "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe), "61"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert out.stdout.count("mismatches 0") == 9, out.stdout


@pytest.mark.parametrize("shape", [(128, 196, 3584, "bf16"), (128, 196, 3584, "f16"), (128, 196, 4096, "f16"), (32, 196, 3584, "bf16"),
                                   (64, 324, 3584, "bf16"), (5, 169, 1024, "bf16"), (2, 512, 1024, "f16"), (6, 7, 1024, "bf16"),
                                   (40, 33, 1024, "bf16"), (130, 48, 1024, "f16"), (448, 16, 1024, "bf16"), (449, 16, 1024, "bf16"),
                                   (512, 196, 3584, "bf16"), (128, 196, 3584, "f32"), (8, 196, 200, "bf16")],
                         ids=lambda s_: "x".join(map(str, s_)))
def test_ord_geometry_tiles_every_frame(shape):
    """Sweep 2's ORD form (round 6: unequal pieces, OrdGeo): host arithmetic only.  Where a shape has the geometry, the
    workgroups' 16-row blocks tile every frame exactly once, in order, every workgroup of a frame has floor or ceil of its
    share, the larger workgroups come first in the launch, and all of them (+ the 64 ORDER riders) fit the 512 resident
    slots; where it has none (fp32, odd widths, more frames than slots) the entry point says 0."""
    import numpy as np
    F, N, D, dn = shape
    L = _ffi.lib()
    cap = 1024
    out = np.zeros((cap, 5), dtype=np.int32)
    n = L.vc2_selftest_ord_pieces(F, N, D, _ffi.DTYPE_CODE[{"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dn]],
                                  out.ctypes.data_as(ctypes.c_void_p), cap)
    assert n >= 0
    if dn == "f32" or D not in (1024, 3584, 4096) or N > 512:
        assert n == 0
        return
    if n == 0:
        assert F > 448                                    # more frames than streaming slots: the row-interleaved sweep
        return
    assert F <= n <= 448
    rows = out[:n]
    nblk = N // 16
    by_frame = {}
    for f, j, sf, b0, nb in rows.tolist():
        by_frame.setdefault(f, []).append((j, sf, b0, nb))
    assert sorted(by_frame) == list(range(F))
    sizes = []
    for f, pcs in by_frame.items():
        pcs.sort()
        sf = pcs[0][1]
        assert [p[0] for p in pcs] == list(range(sf)) and all(p[1] == sf for p in pcs) and 1 <= sf <= 8
        pos = 0
        for _, _, b0, nb in pcs:
            assert b0 == pos and (nb >= 1 or nblk == 0)
            pos += nb
        assert pos == nblk
        assert max(p[3] for p in pcs) - min(p[3] for p in pcs) <= 1
        sizes.append(sf)
    assert max(sizes) - min(sizes) <= 1
    # launch order: the frames with FEWER (= larger) workgroups first
    order = [r[2] for r in rows.tolist()]
    assert order == sorted(order)
