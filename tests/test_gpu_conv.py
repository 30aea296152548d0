"""
GPU parity of the hand-written MFMA convolution unit (aliby_amd/csrc/nn_conv.hip) against a plain PyTorch
fp32 reference of the same op:  conv3x3(relu(scale*x + shift)) + bias + residual.

Two kinds of case: small-integer data, where every product and sum is exact in bf16/fp32, so the comparison
is bit-exact and any mistake in the MFMA fragment / channel-permutation / halo logic shows up as a wrong
integer; and random data, within bf16 output rounding (2^-8 relative to the row's magnitude).
"""

import pytest

pytestmark = pytest.mark.gpu

COMBOS = [(32, 32, False), (32, 64, False), (64, 64, False), (64, 32, True), (64, 64, True), (64, 128, False), (64, 128, True)]


def _run(engine, x, w, scale, shift, bias, res, res_up, in_up, H, W, pool=None):
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    n, cin = x.shape[0], x.shape[-1]
    cout = w.shape[0]
    wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(w), cout, w.shape[1], cin, _ptr(wpk), _stream_ptr()))
    out = torch.full((n, H, W, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_conv3x3_bf16(
        engine.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 1 if shift.ndim == 2 else 0,
        _ptr(bias) if bias is not None else 0, _ptr(res) if res is not None else 0, 1 if res_up else 0, n, H, W, cin, cout,
        1 if in_up else 0, 0, 0, 0, 0, _ptr(pool) if pool is not None else 0, _stream_ptr()))
    torch.cuda.synchronize()
    return out


def _reference(x, w, scale, shift, bias, res, res_up, in_up):
    import torch
    import torch.nn.functional as F

    xf = x.float().permute(0, 3, 1, 2)
    sh = shift if shift.ndim == 2 else shift[None]
    a = torch.relu(xf * scale[None, :, None, None] + sh[:, :, None, None]).bfloat16().float()
    if in_up:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    y = F.conv2d(a, w.bfloat16().float(), None, padding=1)
    if bias is not None:
        y = y + bias[None, :, None, None]
    if res is not None:
        r = res.float().permute(0, 3, 1, 2)
        if res_up:
            r = F.interpolate(r, scale_factor=2, mode="nearest")
        y = y + r
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("shape", [(3, 44, 70), (1, 10, 6), (2, 2, 34), (2, 16, 40)])  # the last: a multiple of the tall tile
@pytest.mark.parametrize("cin,cout,in_up", COMBOS)
def test_conv_unit_exact_on_integer_data(engine, cin, cout, in_up, shape):
    import torch

    torch.backends.cudnn.allow_tf32 = False
    g = torch.Generator().manual_seed(cin * 131 + cout)
    n, H, W = shape  # neither multiples of the tile height nor of the 32-pixel strip; images smaller than one tile
    ih, iw = (H // 2, W // 2) if in_up else (H, W)
    x = torch.randint(-1, 3, (n, ih, iw, cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randint(-1, 2, (cout, cin, 3, 3), generator=g) * (torch.rand(cout, cin, 3, 3, generator=g) < (0.15 if cin < 128 else 0.07))).float().cuda()
    scale = torch.randint(1, 3, (cin,), generator=g).float().cuda()
    shift = torch.randint(-1, 2, (n, cin), generator=g).float().cuda()
    bias = torch.randint(-2, 3, (cout,), generator=g).float().cuda()
    res = torch.randint(-3, 4, (n, H // 2, W // 2, cout), generator=g).to(torch.bfloat16).cuda()
    pool = torch.full((n, H // 2, W // 2, cout), float("nan"), dtype=torch.bfloat16, device="cuda") if cin < 128 else None
    out = _run(engine, x, w, scale, shift, bias, res, True, in_up, H, W, pool=pool)
    ref = _reference(x, w, scale, shift, bias, res, True, in_up)
    assert float(ref.abs().max()) <= 256  # exactly representable in bf16
    assert torch.equal(out.float(), ref)
    # fused max_pool2d(OUT, 2, 2): the next level's input
    if pool is not None:
        ref_pool = torch.nn.functional.max_pool2d(ref.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
        assert torch.equal(pool.float(), ref_pool)


@pytest.mark.parametrize("shape", [(8, 28, 28), (16, 28, 28), (4, 56, 56), (8, 56, 56), (2, 112, 112), (4, 112, 112),
                                   (3, 56, 56), (7, 28, 28)])  # the last two: not a whole group, plain launch
@pytest.mark.parametrize("res_up", [False, True])
@pytest.mark.parametrize("cin,cout", [(64, 128), (64, 64), (32, 64)])
def test_conv_unit_packed_deep_level_images(engine, shape, res_up, cin, cout):
    """28 / 56 / 112-pixel images: G = 224 / W images share a tile row, blocks straddle two images (per-image style
    shift, residual, store and pooled output on either side of the seam)."""
    import torch

    torch.backends.cudnn.allow_tf32 = False
    n, H, W = shape
    g = torch.Generator().manual_seed(n * 1000 + W + cin)
    x = torch.randint(-1, 3, (n, H, W, cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randint(-1, 2, (cout, cin, 3, 3), generator=g) * (torch.rand(cout, cin, 3, 3, generator=g) < 0.15)).float().cuda()
    scale = torch.randint(1, 3, (cin,), generator=g).float().cuda()
    shift = torch.randint(-1, 2, (n, cin), generator=g).float().cuda()
    bias = torch.randint(-2, 3, (cout,), generator=g).float().cuda()
    rh, rw = (H // 2, W // 2) if res_up else (H, W)
    res = torch.randint(-3, 4, (n, rh, rw, cout), generator=g).to(torch.bfloat16).cuda()
    pool = torch.full((n, H // 2, W // 2, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    for use_pool in (None, pool):
        for sh in (shift, shift[0].contiguous()):
            out = _run(engine, x, w, scale, sh, bias, res, res_up, False, H, W, pool=use_pool)
            ref = _reference(x, w, scale, sh, bias, res, res_up, False)
            assert float(ref.abs().max()) <= 256
            assert torch.equal(out.float(), ref)
    ref_pool = torch.nn.functional.max_pool2d(ref.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(pool.float(), ref_pool)


@pytest.mark.parametrize("shape", [(4, 56, 56), (8, 56, 56), (2, 112, 112), (6, 112, 112), (3, 56, 56)])
@pytest.mark.parametrize("cout", [64, 128])
@pytest.mark.parametrize("res_up", [False, True])
def test_conv_unit_packed_with_upsampled_input(engine, shape, cout, res_up):
    """The LDS-DMA variant's packed launch: the half-resolution raw window switches images at the seam."""
    import torch

    torch.backends.cudnn.allow_tf32 = False
    n, H, W = shape
    cin = 64
    g = torch.Generator().manual_seed(n * 77 + W + cout)
    x = torch.randint(-1, 3, (n, H // 2, W // 2, cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randint(-1, 2, (cout, cin, 3, 3), generator=g) * (torch.rand(cout, cin, 3, 3, generator=g) < 0.15)).float().cuda()
    scale = torch.randint(1, 3, (cin,), generator=g).float().cuda()
    shift = torch.randint(-1, 2, (n, cin), generator=g).float().cuda()
    bias = torch.randint(-2, 3, (cout,), generator=g).float().cuda()
    rh, rw = (H // 2, W // 2) if res_up else (H, W)
    res = torch.randint(-3, 4, (n, rh, rw, cout), generator=g).to(torch.bfloat16).cuda()
    for sh in (shift, shift[0].contiguous()):
        for r in (res, None):
            out = _run(engine, x, w, scale, sh, bias, r, res_up, True, H, W)
            ref = _reference(x, w, scale, sh, bias, r, res_up, True)
            assert float(ref.abs().max()) <= 256
            assert torch.equal(out.float(), ref)


@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 44, 70), (1, 10, 6), (2, 64, 32)])
@pytest.mark.parametrize("O", [1, 3])
def test_conv_unit_with_fused_output_head(engine, shape, O):
    """The last unit with the output head in its epilogue == the unit followed by aliby_nn_out_head_bf16, bit for bit,
    with and without writing the unit's own output."""
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    n, H, W = shape
    g = torch.Generator().manual_seed(n * 10 + O)
    x = torch.randn(n, H, W, 32, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.06).float().cuda()
    scale = (torch.rand(32, generator=g) + 0.5).float().cuda()
    shift = (torch.randn(n, 32, generator=g) * 0.2).float().cuda()
    bias = (torch.randn(32, generator=g) * 0.1).float().cuda()
    res = torch.randn(n, H, W, 32, generator=g).to(torch.bfloat16).cuda()
    hs = (torch.rand(32, generator=g) + 0.5).float().cuda()
    hb = (torch.randn(32, generator=g) * 0.2).float().cuda()
    hw = (torch.randn(O, 32, generator=g) * 0.3).float().cuda()
    hbias = torch.randn(O, generator=g).float().cuda()
    out = _run(engine, x, w, scale, shift, bias, res, False, False, H, W)
    want = torch.full((n, O, H, W), float("nan"), device="cuda")
    _lib.check(engine.lib.aliby_nn_out_head_bf16(engine.ctx.handle, _ptr(out), _ptr(hs), _ptr(hb), _ptr(hw), _ptr(hbias), n, H, W, 32, O,
                                                 _ptr(want), _stream_ptr()))
    wpk = torch.empty(32 * 32 * 9, dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(w), 32, 32, 32, _ptr(wpk), _stream_ptr()))
    for keep in (False, True):
        got = torch.full((n, O, H, W), float("nan"), device="cuda")
        out2 = torch.full_like(out, float("nan")) if keep else None
        _lib.check(engine.lib.aliby_nn_conv3x3_head_bf16(
            engine.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out2) if keep else 0, _ptr(scale), _ptr(shift), 32, _ptr(bias), _ptr(res), 0, n, H, W,
            32, 32, _ptr(hs), _ptr(hb), _ptr(hw), _ptr(hbias), O, _ptr(got), _stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        if keep:
            assert torch.equal(out2, out)


@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 44, 70), (1, 10, 6), (5, 14, 30), (2, 16, 62), (1, 2, 2)])
@pytest.mark.parametrize("mode", ["plain", "pool", "head", "head_keep"])
def test_conv_pair_equals_two_launches(engine, shape, mode):
    """The wave-specialised fused pair (aliby_nn_conv3x3_pair_bf16) against two aliby_nn_conv3x3_bf16 launches (and
    aliby_nn_out_head_bf16 for the head variants), bit for bit, on tiles that do and do not divide by 14 x 30."""
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    n, H, W = shape
    g = torch.Generator().manual_seed(n * 31 + H + len(mode))
    x = torch.randn(n, H, W, 32, generator=g).to(torch.bfloat16).cuda()
    pk, ws = [], []
    for k in range(2):
        w = (torch.randn(32, 32, 3, 3, generator=g) * 0.06).float().cuda()
        p = torch.empty(32 * 32 * 9, dtype=torch.bfloat16, device="cuda")
        _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(w), 32, 32, 32, _ptr(p), _stream_ptr()))
        pk.append(p)
        ws.append(w)
    sc = [(torch.rand(32, generator=g) + 0.5).float().cuda() for _ in range(2)]
    sh = [(torch.randn(n, 32, generator=g) * 0.2).float().cuda(), (torch.randn(32, generator=g) * 0.2).float().cuda()]  # per image | shared
    bi = [(torch.randn(32, generator=g) * 0.1).float().cuda() for _ in range(2)]
    res = torch.randn(n, H, W, 32, generator=g).to(torch.bfloat16).cuda()
    pool = mode == "pool" and H % 2 == 0 and W % 2 == 0
    head = mode.startswith("head")
    mid = _run(engine, x, ws[0], sc[0], sh[0], bi[0], None, False, False, H, W)
    pooled2 = torch.full((n, H // 2, W // 2, 32), float("nan"), dtype=torch.bfloat16, device="cuda") if pool else None
    want = _run(engine, mid, ws[1], sc[1], sh[1], bi[1], res, False, False, H, W, pool=pooled2)
    hs, hb = (torch.rand(32, generator=g) + 0.5).float().cuda(), (torch.randn(32, generator=g) * 0.2).float().cuda()
    hw, hbias = (torch.randn(3, 32, generator=g) * 0.3).float().cuda(), torch.randn(3, generator=g).float().cuda()
    want_head = torch.full((n, 3, H, W), float("nan"), device="cuda")
    if head:
        _lib.check(engine.lib.aliby_nn_out_head_bf16(engine.ctx.handle, _ptr(want), _ptr(hs), _ptr(hb), _ptr(hw), _ptr(hbias), n, H, W, 32, 3,
                                                     _ptr(want_head), _stream_ptr()))
    out = torch.full_like(want, float("nan")) if mode != "head" else None
    pooled1 = torch.full_like(pooled2, float("nan")) if pool else None
    got_head = torch.full((n, 3, H, W), float("nan"), device="cuda")
    _lib.check(engine.lib.aliby_nn_conv3x3_pair_bf16(
        engine.ctx.handle, _ptr(x), _ptr(pk[0]), _ptr(pk[1]), _ptr(out) if out is not None else 0, _ptr(sc[0]), _ptr(sh[0]), 32, _ptr(bi[0]),
        _ptr(sc[1]), _ptr(sh[1]), 0, _ptr(bi[1]), _ptr(res), n, H, W, _ptr(pooled1) if pool else 0,
        _ptr(hs) if head else 0, _ptr(hb) if head else 0, _ptr(hw) if head else 0, _ptr(hbias) if head else 0, 3 if head else 0,
        _ptr(got_head) if head else 0, _stream_ptr()))
    torch.cuda.synchronize()
    if out is not None:
        assert torch.equal(out, want)
    if pool:
        assert torch.equal(pooled1, pooled2)
    if head:
        assert torch.equal(got_head, want_head)


@pytest.mark.parametrize("cin,cout,in_up", COMBOS)
def test_conv_unit_random_data_full_tile_shapes(engine, cin, cout, in_up):
    import torch

    g = torch.Generator().manual_seed(7 + cin + cout)
    n = 5
    H = W = 224 if cout == 32 else 112
    ih, iw = (H // 2, W // 2) if in_up else (H, W)
    x = torch.randn(n, ih, iw, cin, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin**0.5)).cuda()
    scale = (torch.rand(cin, generator=g) + 0.5).cuda()
    shift = (torch.randn(cin, generator=g) * 0.2).cuda()  # shared shift (down path)
    res = torch.randn(n, H, W, cout, generator=g).to(torch.bfloat16).cuda()
    out = _run(engine, x, w, scale, shift, None, res, False, in_up, H, W).float()
    ref = _reference(x, w, scale, shift, None, res, False, in_up)
    assert torch.isfinite(out).all()
    err = (out - ref).abs().max() / ref.abs().max()
    assert float(err) < 2.0**-7, float(err)
    # no bias, no residual
    out2 = _run(engine, x, w, scale, shift, None, None, False, in_up, H, W).float()
    ref2 = _reference(x, w, scale, shift, None, None, False, in_up)
    assert float((out2 - ref2).abs().max() / ref2.abs().max()) < 2.0**-7


def test_conv_unit_k_split_over_channel_slices(engine):
    """128 -> 64 through the upsample as two launches over input-channel halves, the second adding to the first."""
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    g = torch.Generator().manual_seed(11)
    n, H, W, ctot, cout = 3, 56, 72, 128, 64
    x = torch.randint(-1, 3, (n, H // 2, W // 2, ctot), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randint(-1, 2, (cout, ctot, 3, 3), generator=g) * (torch.rand(cout, ctot, 3, 3, generator=g) < 0.1)).float().cuda()
    scale = torch.randint(1, 3, (ctot,), generator=g).float().cuda()
    shift = torch.randint(-1, 2, (ctot,), generator=g).float().cuda()
    bias = torch.randint(-2, 3, (cout,), generator=g).float().cuda()
    skip = torch.randint(-3, 4, (n, H, W, cout), generator=g).to(torch.bfloat16).cuda()
    outs = []
    res = skip
    for lo, b in ((0, None), (64, bias)):
        wk = w[:, lo:lo + 64].contiguous()
        wpk = torch.empty(cout * 64 * 9, dtype=torch.bfloat16, device="cuda")
        _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(wk), cout, 64, 64, _ptr(wpk), _stream_ptr()))
        out = torch.empty((n, H, W, cout), dtype=torch.bfloat16, device="cuda")
        _lib.check(engine.lib.aliby_nn_conv3x3_bf16(
            engine.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale[lo:lo + 64]), _ptr(shift[lo:lo + 64]), 0,
            _ptr(b) if b is not None else 0, _ptr(res), 0, n, H, W, 64, cout, 1, ctot, lo, 0, 0, 0, _stream_ptr()))
        res = out
        outs.append(out)
    torch.cuda.synchronize()
    ref = _reference(x, w, scale, shift, bias, skip, False, True)
    assert float(ref.abs().max()) <= 256
    assert torch.equal(outs[-1].float(), ref)


@pytest.mark.parametrize("n", [2, 8])  # 8 images of 28 pixels: the packed launch
def test_conv_unit_n_split_into_output_channel_slices(engine, n):
    """64 -> 256 as two launches writing the two 128-channel halves of one output tensor (residual sliced alike); the 64
    input channels are the upper half of a 128-channel tensor."""
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    g = torch.Generator().manual_seed(12)
    H, W, cin, ctot = 28, 28, 64, 256
    wide = torch.randint(-1, 3, (n, H, W, 2 * cin), generator=g).to(torch.bfloat16).cuda()
    x = wide[..., cin:].contiguous()
    w = (torch.randint(-1, 2, (ctot, cin, 3, 3), generator=g) * (torch.rand(ctot, cin, 3, 3, generator=g) < 0.15)).float().cuda()
    scale = torch.randint(1, 3, (cin,), generator=g).float().cuda()
    shift = torch.randint(-1, 2, (cin,), generator=g).float().cuda()
    bias = torch.randint(-2, 3, (ctot,), generator=g).float().cuda()
    res = torch.randint(-3, 4, (n, H, W, ctot), generator=g).to(torch.bfloat16).cuda()
    out = torch.full((n, H, W, ctot), float("nan"), dtype=torch.bfloat16, device="cuda")
    for n0 in (0, 128):
        wk = w[n0:n0 + 128].contiguous()
        wpk = torch.empty(128 * cin * 9, dtype=torch.bfloat16, device="cuda")
        _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(wk), 128, cin, cin, _ptr(wpk), _stream_ptr()))
        _lib.check(engine.lib.aliby_nn_conv3x3_bf16(
            engine.ctx.handle, _ptr(wide), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 0, _ptr(bias[n0:n0 + 128]), _ptr(res), 0,
            n, H, W, cin, 128, 0, 2 * cin, cin, ctot, n0, 0, _stream_ptr()))
    torch.cuda.synchronize()
    ref = _reference(x, w, scale, shift, bias, res, False, False)
    assert float(ref.abs().max()) <= 256
    assert torch.equal(out.float(), ref)


@pytest.mark.parametrize("cin,pc", [(32, 8), (32, 16), (64, 32)])
def test_conv_unit_with_fused_projection(engine, cin, pc):
    """conv3x3(relu(scale*x + shift)) + bias + conv1x1(x_in) in one launch: the residual block's projection rides in the
    accumulation (exact on integer data)."""
    import torch
    import torch.nn.functional as F
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    g = torch.Generator().manual_seed(cin + pc)
    n, H, W, cout = 2, 37, 45, cin
    x = torch.randint(-1, 3, (n, H, W, cin), generator=g).to(torch.bfloat16).cuda()
    xin = torch.randint(-2, 3, (n, H, W, pc), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randint(-1, 2, (cout, cin, 3, 3), generator=g) * (torch.rand(cout, cin, 3, 3, generator=g) < 0.15)).float().cuda()
    wp = torch.randint(-1, 2, (cout, pc), generator=g).float().cuda()
    scale = torch.randint(1, 3, (cin,), generator=g).float().cuda()
    shift = torch.randint(-1, 2, (cin,), generator=g).float().cuda()
    bias = torch.randint(-2, 3, (cout,), generator=g).float().cuda()
    pc_pad = 16 if cin == 32 else 32
    wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
    ppk = torch.empty(cout * pc_pad, dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(w), cout, cin, cin, _ptr(wpk), _stream_ptr()))
    _lib.check(engine.lib.aliby_nn_pack_conv1x1_bf16(engine.ctx.handle, _ptr(wp), cout, pc, pc_pad, _ptr(ppk), _stream_ptr()))
    out = torch.full((n, H, W, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_conv3x3_proj_bf16(engine.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 0,
                                                     _ptr(bias), n, H, W, cin, cout, _ptr(xin), _ptr(ppk), pc, _stream_ptr()))
    torch.cuda.synchronize()
    ref = _reference(x, w, scale, shift, bias, None, False, False)
    ref = ref + F.conv2d(xin.float().permute(0, 3, 1, 2), wp[:, :, None, None]).permute(0, 2, 3, 1)
    assert float(ref.abs().max()) <= 256
    assert torch.equal(out.float(), ref)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 128), (128, 64), (128, 256), (256, 256), (256, 128)])
def test_conv1x1_exact_on_integer_data(engine, cin, cout):
    import torch
    import torch.nn.functional as F
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    g = torch.Generator().manual_seed(cin * 7 + cout)
    n, H, W = 3, 13, 21  # 819 pixels: not a multiple of the 128 / 64-pixel tile
    x = torch.randint(-2, 3, (n, H, W, cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randint(-1, 2, (cout, cin), generator=g) * (torch.rand(cout, cin, generator=g) < 0.3)).float().cuda()
    bias = torch.randint(-3, 4, (cout,), generator=g).float().cuda()
    wpk = torch.empty(cout * cin, dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_pack_conv1x1_bf16(engine.ctx.handle, _ptr(w), cout, cin, cin, _ptr(wpk), _stream_ptr()))
    out = torch.full((n, H, W, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_conv1x1_bf16(engine.ctx.handle, _ptr(x), _ptr(wpk), _ptr(bias), _ptr(out), n, H, W, cin, cout, _stream_ptr()))
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w[:, :, None, None], bias).permute(0, 2, 3, 1)
    assert float(ref.abs().max()) <= 256
    assert torch.equal(out.float(), ref)


@pytest.mark.parametrize("cin", [1, 2])
def test_first_layer_kernel_matches_torch(engine, cin):
    """float32 NCHW tiles -> c0 = conv3x3(bf16(relu(scale*x + shift))) (bf16 NHWC[32]) + the raw bf16 NHWC[8] copy."""
    import torch
    import torch.nn.functional as F
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    g = torch.Generator().manual_seed(40 + cin)
    n, H, W = 2, 45, 70
    x = torch.randn(n, cin, H, W, generator=g).cuda()
    w = (torch.randn(32, cin, 3, 3, generator=g) * 0.3).bfloat16().float().cuda()
    scale = torch.cat([torch.rand(cin, generator=g) + 0.5, torch.ones(8 - cin)]).cuda()
    shift = torch.cat([torch.randn(cin, generator=g) * 0.2, torch.zeros(8 - cin)]).cuda()
    raw = torch.full((n, H, W, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    c0 = torch.full((n, H, W, 32), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_first_conv_bf16(engine.ctx.handle, _ptr(x), n, cin, H, W, _ptr(scale), _ptr(shift), _ptr(w), _ptr(raw),
                                                   _ptr(c0), _stream_ptr()))
    torch.cuda.synchronize()
    act = torch.relu(x * scale[None, :cin, None, None] + shift[None, :cin, None, None]).bfloat16().float()
    ref = F.conv2d(act, w, None, padding=1).permute(0, 2, 3, 1)
    err = (c0.float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2.0**-7, float(err)
    want_raw = torch.zeros(n, H, W, 8, device="cuda")
    want_raw[..., :cin] = x.permute(0, 2, 3, 1).bfloat16().float()
    assert torch.equal(raw.float(), want_raw)


def test_conv_unit_rejects_unsupported_shapes(engine):
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    x = torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device="cuda")
    f = torch.zeros(16, device="cuda")
    with pytest.raises(Exception, match="unsupported"):
        _lib.check(engine.lib.aliby_nn_conv3x3_bf16(engine.ctx.handle, _ptr(x), _ptr(x), _ptr(x), _ptr(f), _ptr(f), 0, 0, 0, 0,
                                                    1, 8, 8, 16, 16, 0, 0, 0, 0, 0, 0, _stream_ptr()))


# ------------------------------------------------------------------------------------------ the deep-level K-loop kernel
DEEP = [(64, 128, False), (128, 128, False), (128, 256, False), (256, 256, False), (256, 128, True), (128, 128, True)]


def _run_deep(engine, x, w, scale, shift, bias, res, res_up, in_up, H, W, m16=False):
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    n, cin = x.shape[0], x.shape[-1]
    cout = w.shape[0]
    wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
    if m16:  # the 16x16x32 form: its own fragment order
        _lib.check(engine.lib.aliby_nn_pack_conv3x3_deep16_bf16(engine.ctx.handle, _ptr(w.contiguous()), cout, cin, _ptr(wpk), _stream_ptr()))
    else:
        _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(w), cout, w.shape[1], cin, _ptr(wpk), _stream_ptr()))
    out = torch.full((n, H, W, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check((engine.lib.aliby_nn_conv3x3_deep16_bf16 if m16 else engine.lib.aliby_nn_conv3x3_deep_bf16)(
        engine.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 1 if shift.ndim == 2 else 0,
        _ptr(bias) if bias is not None else 0, _ptr(res) if res is not None else 0, 1 if res_up else 0, n, H, W, cin, cout,
        1 if in_up else 0, _stream_ptr()))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("m16", [False, True], ids=["mfma32x32x16", "mfma16x16x32"])
@pytest.mark.parametrize("shape", [(3, 28, 28), (5, 56, 56), (2, 20, 36), (1, 12, 50), (9, 28, 28)])
@pytest.mark.parametrize("cin,cout,in_up", DEEP)
def test_deep_conv_exact_on_integer_data(engine, cin, cout, in_up, shape, m16):
    """aliby_nn_conv3x3_deep_bf16 (one launch, fp32 accumulation over the whole 9*CIN reduction) on data where every product and
    sum is exact: any mistake in the flattened-position / tall-image / fragment logic is a wrong integer.  Images of several
    sizes, batches that end inside a tile, per-sample shifts (a window spans two images), upsampled input and residual."""
    import torch

    g = torch.Generator().manual_seed(cin * 7 + cout + shape[1])
    n, H, W = shape
    ih, iw = (H // 2, W // 2) if in_up else (H, W)
    x = torch.randint(-1, 3, (n, ih, iw, cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randint(-1, 2, (cout, cin, 3, 3), generator=g) * (torch.rand(cout, cin, 3, 3, generator=g) < 0.04)).float().cuda()
    scale = torch.randint(1, 3, (cin,), generator=g).float().cuda()
    shift = torch.randint(-1, 2, (n, cin), generator=g).float().cuda()
    bias = torch.randint(-2, 3, (cout,), generator=g).float().cuda()
    res = torch.randint(-3, 4, (n, H // 2, W // 2, cout), generator=g).to(torch.bfloat16).cuda()
    out = _run_deep(engine, x, w, scale, shift, bias, res, True, in_up, H, W, m16=m16)
    ref = _reference(x, w, scale, shift, bias, res, True, in_up)
    assert float(ref.abs().max()) <= 256
    assert torch.equal(out.float(), ref), (out.float() - ref).abs().max()
    # no residual, shared shift, no bias
    out2 = _run_deep(engine, x, w, scale, shift[0].contiguous(), None, None, False, in_up, H, W, m16=m16)
    assert torch.equal(out2.float(), _reference(x, w, scale, shift[0], None, None, False, in_up))


@pytest.mark.parametrize("cin,cout,in_up", DEEP)
def test_deep_conv_loader_specialised_form_has_the_same_bits(engine, cin, cout, in_up, monkeypatch):
    """ALIBY_DEEP_LS=1 selects k_conv3x3_deep_ls (MFMA waves + LDS-DMA loader waves, not the default: DESIGN.md 3.2): same MFMA
    order, so the same bits as the default kernel on random data."""
    import torch

    g = torch.Generator().manual_seed(cin + cout)
    n, H, W = 5, 56, 56
    ih, iw = (H // 2, W // 2) if in_up else (H, W)
    x = torch.randn(n, ih, iw, cin, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    scale = (torch.rand(cin, generator=g) + 0.5).cuda()
    shift = torch.randn(n, cin, generator=g).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    res = torch.randn(n, H // 2, W // 2, cout, generator=g).to(torch.bfloat16).cuda()
    monkeypatch.delenv("ALIBY_DEEP_LS", raising=False)
    ref = _run_deep(engine, x, w, scale, shift, bias, res, True, in_up, H, W)
    monkeypatch.setenv("ALIBY_DEEP_LS", "1")
    out = _run_deep(engine, x, w, scale, shift, bias, res, True, in_up, H, W)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


def test_deep_conv_random_data_is_closer_to_fp32_than_the_sliced_launches(engine):
    """Random data: the K-loop kernel rounds to bf16 once (|err| <= 2^-8 of the value + fp32 summation noise); the K-split
    launches it replaces rounded their partial sums to bf16 three times for a 256-channel input."""
    import torch

    g = torch.Generator().manual_seed(5)
    n, H, W, cin, cout = 4, 28, 28, 256, 256
    x = torch.randn((n, H, W, cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (3 * cin**0.5)).float().cuda()
    scale = (1 + 0.1 * torch.randn(cin, generator=g)).float().cuda()
    shift = (0.1 * torch.randn((n, cin), generator=g)).float().cuda()
    bias = torch.randn(cout, generator=g).float().cuda()
    res = torch.randn((n, H, W, cout), generator=g).to(torch.bfloat16).cuda()
    out = _run_deep(engine, x, w, scale, shift, bias, res, False, False, H, W).float()
    ref = _reference(x, w, scale, shift, bias, res, False, False)
    err = (out - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 2e-3).all())  # one bf16 rounding of the result + fp32 summation noise
    rel_l2 = float((out - ref).norm() / ref.norm())
    assert rel_l2 < 2.5e-3, rel_l2  # bf16 output rounding alone is ~1.6e-3 rms
    # the slice launches: 4 K-slices x 2 N-slices chained through the bf16 output
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    sliced = torch.empty((n, H, W, cout), dtype=torch.bfloat16, device="cuda")
    for n0 in range(0, cout, 128):
        cur = res
        for k0 in range(0, cin, 64):
            wk = w[n0:n0 + 128, k0:k0 + 64].contiguous()
            wpk = torch.empty(128 * 64 * 9, dtype=torch.bfloat16, device="cuda")
            _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(wk), 128, 64, 64, _ptr(wpk), _stream_ptr()))
            last = k0 + 64 == cin
            _lib.check(engine.lib.aliby_nn_conv3x3_bf16(
                engine.ctx.handle, _ptr(x), _ptr(wpk), _ptr(sliced), _ptr(scale[k0:k0 + 64].contiguous()), _ptr(shift[:, k0:k0 + 64].contiguous()), 1,
                _ptr(bias[n0:n0 + 128].contiguous()) if last else 0, _ptr(cur), 0, n, H, W, 64, 128, 0, cin, k0, cout, n0, 0, _stream_ptr()))
            cur = sliced
    torch.cuda.synchronize()
    rel_sliced = float((sliced.float() - ref).norm() / ref.norm())
    print(f"rel-L2 vs fp32 reference: K-loop {rel_l2:.2e}, K/N-slice launches {rel_sliced:.2e}")
    assert rel_l2 < rel_sliced


def test_maxpool2_bf16(engine):
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    x = torch.randn((3, 12, 20, 128), generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()
    out = torch.empty((3, 6, 10, 128), dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_maxpool2_bf16(engine.ctx.handle, _ptr(x), _ptr(out), 3, 12, 20, 128, _stream_ptr()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.max_pool2d(x.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(out.float(), ref)


@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 44, 70), (1, 10, 6), (5, 14, 30), (2, 16, 62), (1, 2, 2), (2, 31, 33)])
@pytest.mark.parametrize("cin", [2, 1])
def test_first_pair_equals_first_conv_then_projected_unit(engine, shape, cin):
    """aliby_nn_first_pair_bf16 (round 3: the first layer as the producer of the second unit, neither c0 nor the raw bf16 copy
    in HBM) against the two launches it replaces — aliby_nn_first_conv_bf16 then aliby_nn_conv3x3_proj_bf16 — bit for bit, on
    tiles that do and do not divide by 14 x 30, one and two input channels."""
    import torch
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    n, H, W = shape
    g = torch.Generator().manual_seed(n * 17 + H + cin)
    tiles = (torch.randn(n, cin, H, W, generator=g) * 2.0).float().cuda().contiguous()
    s0 = torch.zeros(8).float()
    h0 = torch.zeros(8).float()
    s0[:cin] = torch.rand(cin, generator=g) + 0.5
    h0[:cin] = torch.randn(cin, generator=g) * 0.3
    s0, h0 = s0.cuda(), h0.cuda()
    w0 = (torch.randn(32, cin, 3, 3, generator=g) * 0.3).to(torch.bfloat16).float().cuda().contiguous()  # bf16-representable
    w1 = (torch.randn(32, 32, 3, 3, generator=g) * 0.06).float().cuda()
    wpk1 = torch.empty(32 * 32 * 9, dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_pack_conv3x3_bf16(engine.ctx.handle, _ptr(w1), 32, 32, 32, _ptr(wpk1), _stream_ptr()))
    s1 = (torch.rand(32, generator=g) + 0.5).float().cuda()
    h1 = (torch.randn(32, generator=g) * 0.2).float().cuda()
    b1 = (torch.randn(32, generator=g) * 0.1).float().cuda()
    wp = torch.zeros(32, 8)
    wp[:, :cin] = torch.randn(32, cin, generator=g) * 0.4
    wp = wp.float().cuda().contiguous()
    ppk = torch.empty(32 * 16, dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_pack_conv1x1_bf16(engine.ctx.handle, _ptr(wp), 32, 8, 16, _ptr(ppk), _stream_ptr()))
    raw = torch.full((n, H, W, 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    c0 = torch.full((n, H, W, 32), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_first_conv_bf16(engine.ctx.handle, _ptr(tiles), n, cin, H, W, _ptr(s0), _ptr(h0), _ptr(w0), _ptr(raw), _ptr(c0),
                                                   _stream_ptr()))
    want = torch.full((n, H, W, 32), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(engine.lib.aliby_nn_conv3x3_proj_bf16(engine.ctx.handle, _ptr(c0), _ptr(wpk1), _ptr(want), _ptr(s1), _ptr(h1), 0, _ptr(b1), n, H, W,
                                                     32, 32, _ptr(raw), _ptr(ppk), 8, _stream_ptr()))
    got = torch.full_like(want, float("nan"))
    _lib.check(engine.lib.aliby_nn_first_pair_bf16(engine.ctx.handle, _ptr(tiles), n, cin, H, W, _ptr(s0), _ptr(h0), _ptr(w0), _ptr(wpk1), _ptr(s1),
                                                   _ptr(h1), _ptr(b1), _ptr(ppk), _ptr(got), _stream_ptr()))
    torch.cuda.synchronize()
    assert not torch.isnan(want.float()).any()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
