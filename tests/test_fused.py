"""f1: vc2_gather_scatter / vc2_keep_positions / vc2_compress_tail against their torch formulations
(the statements they replace in the hooks: vidcom2.py:91, models/llava.py:160-168, models/qwen2_5_vl.py:153-182,
models/qwen3_vl.py:140-165).  Pure copies and index arithmetic: bit-exact."""
import pytest
import torch
import torch.nn.functional as F

from vidcom2_amd import synth

pytestmark = pytest.mark.gpu
DTS = [torch.bfloat16, torch.float16, torch.float32]


def _rand(rows, D, dt, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(rows, D, generator=g).to(dt).cuda()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("D", [64, 3584, 100, 7])          # 7 / 100: rows that are not multiples of 16 bytes
def test_gather_scatter_matches_indexing(dt, D):
    from vidcom2_amd.fused import gather_scatter
    g = torch.Generator().manual_seed(D)
    srcs = [_rand(300, D, dt, 10 + t) for t in range(3)]
    idx = torch.randint(0, 300, (123,), generator=g).cuda()
    out = gather_scatter(srcs, idx)
    for s, o in zip(srcs, out):
        assert torch.equal(o, s[idx])
    # tail rows land behind the gathered rows of the FIRST tensor only
    tail = _rand(2, D, dt, 99)
    out = gather_scatter(srcs, idx, tail=tail)
    assert torch.equal(out[0], torch.cat((srcs[0][idx], tail))) and torch.equal(out[1], srcs[1][idx])
    # scatter to given positions of a caller-provided buffer; untouched rows stay as they were
    pos = torch.randperm(400, generator=g)[:123].cuda()
    dsts = [torch.full((400, D), 3.0, dtype=dt, device="cuda") for _ in srcs]
    gather_scatter(srcs, idx, dst_pos=pos, dsts=dsts)
    for s, d in zip(srcs, dsts):
        want = torch.full((400, D), 3.0, dtype=dt, device="cuda")
        want[pos] = s[idx]
        assert torch.equal(d, want)


def test_gather_scatter_device_count_identity_and_status():
    from vidcom2_amd.fused import gather_scatter
    src = _rand(50, 32, torch.bfloat16, 1)
    n_dev = torch.tensor([17], dtype=torch.int64, device="cuda")
    tail = _rand(1, 32, torch.bfloat16, 2)
    dst = torch.zeros((41, 32), dtype=torch.bfloat16, device="cuda")
    gather_scatter([src], None, n=40, n_dev=n_dev, dsts=[dst], tail=tail)          # idx None: rows 0..n-1
    assert torch.equal(dst[:17], src[:17]) and torch.equal(dst[17], tail[0]) and not dst[18:].any()
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    bad = torch.tensor([0, 50, 3, -1], dtype=torch.int64, device="cuda")
    out = torch.zeros((4, 32), dtype=torch.bfloat16, device="cuda")
    gather_scatter([src], bad, dsts=[out], status=status)
    assert status.item() == 2 and torch.equal(out[0], src[0]) and torch.equal(out[2], src[3]) and not out[1].any()
    with pytest.raises(ValueError):
        gather_scatter([src] * 9)
    with pytest.raises(RuntimeError):
        gather_scatter([src, _rand(50, 16, torch.bfloat16, 3)])
    with pytest.raises(RuntimeError, match="CPU"):
        gather_scatter([src.cpu()])


@pytest.mark.parametrize("S,seed", [(1, 0), (37, 1), (1024, 2), (1025, 3), (20000, 4), (131072, 5)])
def test_keep_positions_matches_torch(S, seed):
    from vidcom2_amd.fused import keep_positions
    g = torch.Generator().manual_seed(seed)
    vm = torch.rand(S, generator=g) < 0.7
    im = (~vm) & (torch.rand(S, generator=g) < 0.2)                 # image tokens: visual but not video
    n_video = int(vm.sum())
    K = n_video // 3
    kept = torch.sort(torch.randperm(max(n_video, 1), generator=g)[:K]).values
    flags = ~vm
    flags[vm.nonzero().squeeze(-1)[kept]] = True
    want_keep = flags.nonzero().squeeze(-1)
    vis = vm | im
    want_rows = flags[vis].nonzero().squeeze(-1)
    keep, rows = keep_positions(vm.cuda(), kept.cuda(), n_video, vis.cuda(), int(vis.sum()))
    assert torch.equal(keep.cpu(), want_keep) and torch.equal(rows.cpu(), want_rows)
    keep2, rows2 = keep_positions(vm.cuda()[None], kept.cuda())                  # counts taken on the device
    assert torch.equal(keep2.cpu(), want_keep) and rows2 is None


def test_keep_positions_and_gather_report_bad_indices():
    """What the fused ops replace raises on a bad index (torch indexing); so do they: the kernels never write past
    the buffers they were given and the host raises instead of handing out uninitialised rows."""
    from vidcom2_amd.fused import gather_scatter, keep_positions
    vm = (torch.arange(200) % 3 != 0).cuda()                        # 133 video positions
    n_video = int(vm.sum())
    kept = torch.arange(0, 40, 2).cuda()
    keep, _ = keep_positions(vm, kept, n_video)                     # sound call
    assert keep.numel() == 200 - n_video + 20 and int(keep.min()) >= 0
    with pytest.raises(IndexError, match="video positions"):
        keep_positions(vm, kept, n_video + 7)                        # the caller's count is too large ...
    with pytest.raises(IndexError, match="video positions"):
        keep_positions(vm, kept, n_video - 5)                        # ... or too small: nothing is written past keep
    with pytest.raises(IndexError, match="ascending"):
        keep_positions(vm, torch.tensor([3, 3, 9]).cuda(), n_video)  # duplicates
    with pytest.raises(IndexError, match="ascending"):
        keep_positions(vm, torch.tensor([9, 3]).cuda(), n_video)     # unsorted
    with pytest.raises(IndexError, match="ascending"):
        keep_positions(vm, torch.tensor([5, n_video]).cuda(), n_video)   # ordinal out of range
    src = _rand(50, 32, torch.bfloat16, 4)
    with pytest.raises(IndexError, match="out of range"):
        gather_scatter([src], torch.tensor([0, 50, 3]).cuda(), check=True)
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    gather_scatter([src], torch.tensor([0, -1, 3]).cuda(), status=st)   # caller-owned status word: reported there
    assert int(st.item()) & 2


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_compress_with_tail_equals_compress_then_cat(dt):
    import vidcom2_amd as V
    x = synth.make(6, 196, 256, dt, 5).reshape(-1, 256).cuda()
    nl = _rand(1, 256, dt, 8)
    a = V.compress(x, 196, 0.25)
    b = V.compress(x, 196, 0.25, tail=nl)
    assert b.K == a.K and b.rows.shape[0] == a.K + 1
    assert torch.equal(b.rows, torch.cat((a.rows, nl))) and torch.equal(a.global_idx, b.global_idx)


def test_newline_fusion_returns_the_gather_buffer():
    """The LLaVA one_token branch: rows.flatten -> torch.cat((rows, newline[None])) costs no second copy."""
    from vidcom2_amd.models.llava import _CompressOnFlatten
    import vidcom2_amd as V
    D = 128
    feats = synth.make(4, 196, D, torch.bfloat16, 3).reshape(4, 196, D).cuda()
    nl = _rand(1, D, torch.bfloat16, 4)[0]
    t = feats.as_subclass(_CompressOnFlatten)
    t._vc2_newline = nl
    rows = t.flatten(0, 1)
    out = torch.cat((rows, nl[None].to(rows.device)), dim=0)
    assert type(out) is torch.Tensor and out.data_ptr() == rows.data_ptr() and out.shape[0] == rows.shape[0] + 1
    want = V.vidcom2_compression(feats.flatten(0, 1))
    assert torch.equal(out, torch.cat((want, nl[None])))
    # any other use behaves like a plain tensor
    other = _rand(1, D, torch.bfloat16, 5)
    o2 = torch.cat((rows, other), dim=0)
    assert type(o2) is torch.Tensor and torch.equal(o2, torch.cat((want, other)))
    assert type(rows + 1) is torch.Tensor and type(rows[:3]) is torch.Tensor


# ---- f3: get_2dPool fused with sweep 1 ------------------------------------------------------------------------
def _torch_pool(x, H, W, mode):
    """What LLaVA's get_2dPool computes (llava_arch.py:171-190), on the CPU with torch's own kernels."""
    import math
    import torch.nn.functional as Fn
    F, _, D = x.shape
    t = x.cpu().view(F, H, W, D).permute(0, 3, 1, 2).contiguous()
    if mode == "average":
        t = Fn.avg_pool2d(t, 2)
    elif mode == "max":
        t = Fn.max_pool2d(t, 2)
    else:
        t = Fn.interpolate(t, size=[math.ceil(H / 2), math.ceil(W / 2)], mode="bilinear")
    return t.permute(0, 2, 3, 1).reshape(F, -1, D)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("mode", ["average", "max"])
@pytest.mark.parametrize("H,D", [(27, 256), (28, 3584), (6, 100)])
def test_pool_stats_bits_match_torch_pooling(dt, mode, H, D):
    from vidcom2_amd.fused import pool_stats
    g = torch.Generator().manual_seed(H + D)
    x = torch.randn(5, H * H, D, generator=g).to(dt)
    if mode == "max":
        x[0, 3, 5] = float("nan")
    out, ws = pool_stats(x.cuda(), H, H, mode)
    want = _torch_pool(x, H, H, mode)
    assert out.shape == want.shape
    assert torch.equal(out.cpu().view(torch.int16 if dt != torch.float32 else torch.int32),
                       want.view(torch.int16 if dt != torch.float32 else torch.int32))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("H,W,D", [(27, 27, 3584), (27, 27, 256), (24, 24, 64), (13, 27, 48), (9, 9, 16)])
def test_pool_stats_bilinear_bits_match_torch_interpolate(dt, H, W, D):
    """LLaVA-OneVision's pooling: interpolate(size=ceil/2, mode="bilinear") on the CPU, bit for bit (fp32 included:
    the order in which ATen's vector loop adds its four products was identified against torch itself)."""
    from vidcom2_amd.fused import pool_stats
    g = torch.Generator().manual_seed(7 + D)
    F = 2 if D > 1000 else 4
    x = torch.randn(F, H * W, D, generator=g).to(dt)
    out, _ = pool_stats(x.cuda(), H, W, "bilinear")
    want = _torch_pool(x, H, W, "bilinear")
    assert out.shape == want.shape == (F, -(-H // 2) * -(-W // 2), D)
    it = torch.int32 if dt == torch.float32 else torch.int16
    assert torch.equal(out.cpu().view(it), want.view(it))


def test_pool_stats_bilinear_refuses_feature_sizes_off_torchs_vector_width():
    from vidcom2_amd.fused import pool_stats
    with pytest.raises(NotImplementedError):
        pool_stats(torch.zeros(1, 81, 24, dtype=torch.bfloat16, device="cuda"), 9, 9, "bilinear")     # 24 % 16 != 0
    with pytest.raises(NotImplementedError):
        pool_stats(torch.zeros(1, 81, 12, dtype=torch.float32, device="cuda"), 9, 9, "bilinear")      # 12 % 8 != 0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("mode,H", [("average", 28), ("max", 28), ("bilinear", 27)])
def test_pass_with_fused_stats_equals_pass_on_the_pooled_tensor(dt, mode, H):
    import vidcom2_amd as V
    from vidcom2_amd.fused import pool_stats
    D = 512
    x = synth.make(8, H * H, D, dt, 11).reshape(8, H * H, D).cuda()
    pooled, ws = pool_stats(x, H, H, mode)
    assert pooled.shape[1] == 196
    flat = pooled.flatten(0, 1)
    a = V.compress(flat, 196, 0.25, want_scores=True)
    b = V.compress(flat, 196, 0.25, want_scores=True, stats_ws=ws)
    assert a.K == b.K and torch.equal(a.global_idx, b.global_idx) and torch.equal(a.ks, b.ks)
    assert torch.equal(a.rows, b.rows) and torch.equal(a.v_score, b.v_score) and torch.equal(a.f_score, b.f_score)


@pytest.mark.parametrize("newline,side,mode", [("one_token", 28, "average"), ("grid", 26, "average"),
                                               ("grid", 27, "max"), ("one_token", 27, "bilinear")])
def test_llava_hook_with_fused_pooling_equals_unfused(newline, side, mode, monkeypatch):
    """The hook with get_2dPool + sweep 1 fused gives the same prompt embeddings as with the model's own pooling."""
    import types
    import _stub_models as S
    from vidcom2_amd.models.llava import cus_prepare_inputs_labels_for_multimodal as hook
    D, F = 256, 6

    class Pooling(S.StubLlava):
        def get_2dPool(self, image_feature, stride=2):
            return _torch_pool(image_feature, side, side, self.config.mm_spatial_pool_mode).to(image_feature.device)

    def run(knob):
        monkeypatch.setenv("VC2_FUSED_POOL", knob)
        monkeypatch.setenv("R_RATIO", "0.25")
        m = Pooling(D, torch.bfloat16, "cuda", 3, "spatial_unpad", newline)
        m.config.mm_spatial_pool_mode = mode
        m.model._tower.num_patches_per_side = side
        feats = synth.make(F, side * side, D, torch.bfloat16, 21).reshape(F, side * side, D, 1).cuda()
        ids = torch.tensor([[5, 6, S.IMAGE_TOKEN_INDEX, 7]], device="cuda")
        m.prepare_inputs_labels_for_multimodal = types.MethodType(hook, m)
        return m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats], ["video"], None)[4]

    if newline == "grid":
        # llava_vid frames are 13 x 13 = 169 tokens
        assert side // 2 == 13
    fused, plain = run(""), run("off")
    assert fused.shape == plain.shape and torch.equal(fused, plain)


def test_pool_stats_bilinear_refuses_grids_torch_pools_with_another_kernel():
    """Pinned torch behaviour (2.10, x86): an NCHW-contiguous input goes through the vectorised channels-last bilinear
    kernel -- whose association vc2_pool_stats reproduces -- only while out_h + out_w <= 128."""
    from vidcom2_amd.fused import pool_stats
    a = torch.randn(1, 16, 127, 127)
    same = F.interpolate(a, size=[64, 64], mode="bilinear")
    assert torch.equal(same, F.interpolate(a.contiguous(memory_format=torch.channels_last), size=[64, 64], mode="bilinear"))
    b = torch.randn(1, 16, 129, 129)
    assert not torch.equal(F.interpolate(b, size=[65, 65], mode="bilinear"),
                           F.interpolate(b.contiguous(memory_format=torch.channels_last), size=[65, 65], mode="bilinear"))
    x = torch.zeros(1, 129 * 129, 16, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(NotImplementedError, match="128"):
        pool_stats(x, 129, 129, "bilinear")
    pool_stats(x, 129, 129, "average")                                # (the other modes have no such switch)
