"""-m gpu: the bf16 MFMA convolution (tpspp_conv2d_bf16_fwd) against PyTorch-CPU.

bf16 x bf16 products are exact in fp32, so with the operands rounded to bf16 on both sides the only
differences are the fp32 summation order (fp32 outputs: 1e-5 of the tensor's scale) and, for bf16
outputs, an occasional flip of the final rounding (one bf16 ulp = 2^-8 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tps_pp_amd import ops, synth

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rb(x):
    """round to bf16, keep fp32 storage"""
    return x.to(torch.bfloat16).float()


def ref_conv(srcs, w, b, k, stride, relu, res, res_mode):
    xs = []
    for x, uh, uw in srcs:
        x = rb(x).double()
        xs.append(F.interpolate(x, scale_factor=(uh, uw), mode="nearest") if (uh, uw) != (1, 1) else x)
    y = F.conv2d(torch.cat(xs, 1), rb(w).double(), None if b is None else b.double(), stride=stride,
                 padding=(k - 1) // 2)
    if res_mode == 2:
        y = y + res.double()
    if relu:
        y = F.relu(y)
    if res_mode == 1:
        y = y + res.double()
    return y.float()


CASES = [
    # name, sources [(C,H,W,uh,uw,dtype)], Cout, k, stride, relu, res_mode, res dtype, N
    ("down0 1x1 32->64 @32x128", [(32, 32, 128, 1, 1, "bf16")], 64, 1, (1, 1), True, 0, None, 3),
    ("down0_1 3x3 s2 64->64 @32x128", [(64, 32, 128, 1, 1, "bf16")], 64, 3, (2, 2), True, 0, None, 2),
    ("down_feat 1x1 cat(64,64,up 64)->64", [(64, 32, 128, 1, 1, "bf16"), (64, 32, 128, 1, 1, "bf16"),
                                            (64, 16, 64, 2, 2, "bf16")], 64, 1, (1, 1), True, 0, None, 2),
    ("k_encoder.0 3x3 192->64 @16x64", [(64, 16, 64, 1, 1, "bf16")] * 3, 64, 3, (1, 1), True, 0, None, 2),
    ("k_encoder.1 3x3 s2 @16x64 -> 8x32", [(64, 16, 64, 1, 1, "bf16")], 64, 3, (2, 2), True, 0, None, 3),
    ("k_encoder.2 3x3 s2 @8x32 -> 4x16", [(64, 8, 32, 1, 1, "bf16")], 64, 3, (2, 2), True, 0, None, 5),
    ("k_encoder.3 3x3 s(2,1) @4x16 -> 2x16", [(64, 4, 16, 1, 1, "bf16")], 64, 3, (2, 1), True, 0, None, 9),
    ("k_decoder.0 up(2,1)+3x3 + skip, f32 source", [(64, 2, 16, 2, 1, "f32")], 64, 3, (1, 1), True, 1, "bf16", 5),
    ("k_decoder.1 up2+3x3 + skip @8x32", [(64, 4, 16, 2, 2, "bf16")], 64, 3, (1, 1), True, 1, "bf16", 3),
    ("k_decoder.2 up2+3x3 + skip @16x64", [(64, 8, 32, 2, 2, "bf16")], 64, 3, (1, 1), True, 1, "f32", 2),
    ("stem 3x3 3->32 @32x128 f32 image", [(3, 32, 128, 1, 1, "f32")], 32, 3, (1, 1), True, 0, None, 2),
    ("BasicBlock conv2 3x3 s2 + residual before relu", [(64, 32, 128, 1, 1, "bf16")], 128, 3, (2, 2), True, 2, "bf16", 2),
    ("downsample 1x1 s2 64->128 @32x128", [(64, 32, 128, 1, 1, "bf16")], 128, 1, (2, 2), False, 0, None, 3),
    ("downsample 1x1 s2 ragged 40->72 @9x13", [(40, 9, 13, 1, 1, "bf16")], 72, 1, (2, 2), False, 0, None, 2),
    ("ragged 3x3 16->40 @9x13", [(16, 9, 13, 1, 1, "bf16")], 40, 3, (1, 1), True, 0, None, 3),
    ("ragged 1x1 40->132 @12x44", [(40, 12, 44, 1, 1, "f32")], 132, 1, (1, 1), False, 2, "f32", 3),
    ("1x1 256->512 @4x16", [(256, 4, 16, 1, 1, "bf16")], 512, 1, (1, 1), True, 0, None, 7),
    ("3x3 @16x16 odd channels 24->64", [(24, 16, 16, 1, 1, "bf16")], 64, 3, (1, 1), False, 0, None, 2),
    # at most 32 output channels (round 6: the one-accumulator form of the tiled kernel in the three-term split)
    ("BasicBlock 3x3 32->32 + residual @16x64", [(32, 16, 64, 1, 1, "f32")], 32, 3, (1, 1), True, 2, "f32", 3),
    ("3x3 s2 32->32 + residual @32x128 -> 16x64", [(32, 32, 128, 1, 1, "f32")], 32, 3, (2, 2), True, 2, "f32", 2),
    ("1x1 32->24 ragged @16x64", [(32, 16, 64, 1, 1, "f32")], 24, 1, (1, 1), True, 0, None, 3),
]


@pytest.mark.parametrize("out_dtype", ["bf16", "f32"])
@pytest.mark.parametrize("name,srcs,cout,k,stride,relu,res_mode,res_dt,N", CASES, ids=[c[0] for c in CASES])
def test_conv_bf16_matches_cpu_reference(cuda, name, srcs, cout, k, stride, relu, res_mode, res_dt, N, out_dtype):
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}
    xs = [(t(synth.dyadic((N, c, h, w), f"{name}.x{i}", 1)), uh, uw) for i, (c, h, w, uh, uw, _) in enumerate(srcs)]
    cin = sum(s_[0] for s_ in srcs)
    w = t(synth.dyadic((cout, cin, k, k), name + ".w", 1, 1.0 / np.sqrt(cin * k * k)))
    b = t(synth.dyadic((cout,), name + ".b", 1, 0.1))
    y0 = ref_conv(xs, w, b, k, stride, relu, None, 0)
    res = None
    if res_mode:
        res = t(synth.dyadic(tuple(y0.shape), name + ".r", 1))
        if res_dt == "bf16":
            res = rb(res)
    ref = ref_conv(xs, w, b, k, stride, relu, res, res_mode)
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda))
    dsrcs = [(x.to(cuda).to(dt[s_[5]]), uh, uw) for (x, uh, uw), s_ in zip(xs, srcs)]
    got = ops.conv2d_bf16(dsrcs, cw, stride, relu, None if res is None else res.to(cuda).to(dt[res_dt]),
                          res_mode, out_dtype=dt[out_dtype])
    assert got.dtype == dt[out_dtype] and tuple(got.shape) == tuple(ref.shape)
    got = got.float().cpu()
    scale = float(ref.abs().max())
    if out_dtype == "f32":
        assert float((got - ref).abs().max()) <= 1e-5 * scale, name
    else:
        # within one bf16 ulp of the exactly rounded value
        err = (got - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-6 * scale).all()), (name, float(err.max()))
        # and mostly the exactly rounded value itself
        assert float((got == rb(ref)).float().mean()) > 0.98, name


def test_conv_bf16_empty_batch_and_single_image(cuda):
    w = t(synth.dyadic((64, 32, 3, 3), "eb.w", 1, 0.1))
    for x3 in (False, True):
        cw = ops.prep_conv_weight_bf16(w.to(cuda), x3=x3)
        e = ops.conv2d_bf16([torch.zeros((0, 32, 8, 8), device=cuda)], cw, 1, True)
        assert tuple(e.shape) == (0, 64, 8, 8)
        x = t(synth.dyadic((1, 32, 5, 7), "eb.x", 1))
        got = ops.conv2d_bf16([x.to(cuda)], cw, 2, False, out_dtype=torch.float32).cpu()
        ref = F.conv2d((x if x3 else rb(x)).double(), (w if x3 else rb(w)).double(), stride=2, padding=1).float()
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_conv_bf16_argument_errors(cuda):
    from tps_pp_amd import _lib
    w = torch.zeros((64, 32, 3, 3), device=cuda)
    cw = ops.prep_conv_weight_bf16(w)
    with pytest.raises(ValueError):
        ops.conv2d_bf16([torch.zeros((1, 16, 8, 8), device=cuda)], cw)          # Cin mismatch
    with pytest.raises(TypeError):
        ops.conv2d_bf16([torch.zeros((1, 32, 8, 8), device=cuda, dtype=torch.float16)], cw)
    with pytest.raises(_lib.TpsppError):
        ops.conv2d_bf16([torch.zeros((1, 32, 8, 8))], cw)                       # CPU tensor
    with pytest.raises(_lib.TpsppError):
        ops.conv2d_bf16([torch.zeros((1, 32, 8, 8), device=cuda)], cw, stride=(1, 2))   # no such kernel


BLK_CASES = [c for c in CASES if all(s_[0] % 8 == 0 for s_ in c[1]) and c[2] % 8 == 0]


@pytest.mark.parametrize("name,srcs,cout,k,stride,relu,res_mode,res_dt,N", BLK_CASES, ids=[c[0] for c in BLK_CASES])
def test_conv_bf16x3_fp32_blocked_layout_is_the_same_arithmetic(cuda, name, srcs, cout, k, stride, relu, res_mode, res_dt, N):
    """Layout code 3 (ops.Blocked32: (N, C/8, H, W, 8) float32) with the three-term split: the same bits as the fp32 NCHW
    call for blocked sources / residual / output and for a mix of blocked and NCHW sources."""
    xs = [(t(synth.dyadic((N, c, h, w), f"{name}.x{i}", 1)).to(cuda), uh, uw) for i, (c, h, w, uh, uw, _) in enumerate(srcs)]
    cin = sum(s_[0] for s_ in srcs)
    w = t(synth.dyadic((cout, cin, k, k), name + ".w", 1, 1.0 / np.sqrt(cin * k * k)))
    b = t(synth.dyadic((cout,), name + ".b", 1, 0.1))
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda), x3=True)
    f32 = torch.float32
    plain = ops.conv2d_bf16(xs, cw, stride, relu=relu, out_dtype=f32)
    res = t(synth.dyadic(tuple(plain.shape), name + ".r", 1)).to(cuda) if res_mode else None
    want = ops.conv2d_bf16(xs, cw, stride, relu=relu, residual=res, res_mode=res_mode, out_dtype=f32)
    bsrc = [(ops.Blocked32.from_nchw(x), uh, uw) for x, uh, uw in xs]
    bres = ops.Blocked32.from_nchw(res) if res is not None else None
    got = ops.conv2d_bf16(bsrc, cw, stride, relu=relu, residual=bres, res_mode=res_mode, out_dtype=f32, out_blocked=True)
    assert isinstance(got, ops.Blocked32) and got.shape == tuple(want.shape)
    assert torch.equal(got.nchw().view(torch.int32), want.view(torch.int32)), "blocked in / blocked out"
    got2 = ops.conv2d_bf16([bsrc[0]] + xs[1:], cw, stride, relu=relu, residual=res, res_mode=res_mode, out_dtype=f32)
    assert torch.equal(got2.view(torch.int32), want.view(torch.int32)), "mixed sources, NCHW residual and output"
    with pytest.raises(ValueError):                        # bf16 blocked maps do not go with x3 weights, nor Blocked32 without
        ops.conv2d_bf16([(ops.Blocked.from_nchw(xs[0][0]), xs[0][1], xs[0][2])] + xs[1:], cw, stride)
    with pytest.raises(ValueError):
        ops.conv2d_bf16(bsrc, ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda)), stride)


PERSIST_CASES = [
    # name, sources [(C,H,W,uh,uw)], stride, res_mode, fp32 output, N   (tiles: 2 per 16x64 image, 8 per stride-2 image;
    # a workgroup takes two tiles per trip, 256 workgroups)
    ("dec3 64->64 @16x64, fp32 out, 3 trips", [(64, 16, 64, 1, 1)], (1, 1), 0, True, 601),
    ("dec2 up2 + skip @16x64, 2 trips, ragged", [(64, 8, 32, 2, 2)], (1, 1), 1, False, 301),
    ("enc0 192->64 @16x64 (streamed weight), 2 trips", [(64, 16, 64, 1, 1)] * 3, (1, 1), 0, False, 290),
    ("down0_1 s2 64->64 @32x128, 2 trips, ragged", [(64, 32, 128, 1, 1)], (2, 2), 0, False, 67),
    ("two chunks only 32->64 @16x64", [(32, 16, 64, 1, 1)], (1, 1), 2, False, 5),
    ("one tile", [(64, 8, 64, 1, 1)], (1, 1), 0, False, 1),
    ("dec1 up2 + skip @8x32 (8x32 tiles)", [(64, 4, 16, 2, 2)], (1, 1), 1, False, 515),
    ("enc1 s2 @16x64 -> 8x32 (4x32 tiles)", [(64, 16, 64, 1, 1)], (2, 2), 0, False, 259),
]


@pytest.mark.parametrize("name,srcs,stride,res_mode,f32_out,N", PERSIST_CASES, ids=[c[0] for c in PERSIST_CASES])
def test_conv_bf16_persistent_kernel_is_the_tiled_kernel_bit_for_bit(cuda, name, srcs, stride, res_mode, f32_out, N):
    """tpspp_conv_bf16_persist.hip (persistent workgroups, two teams per workgroup, LDS-DMA patch) against the tiled kernel on the
    same blocked tensors (tpspp_conv_set_tuning bit 1 switches it off): several trips per workgroup, a last trip that only
    some workgroups make, upsampled and concatenated sources, blocked residual, both output forms."""
    from tps_pp_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(len(name))
    xs = [(ops.Blocked.from_nchw(torch.randn((N, c, h, w), generator=g).to(cuda)), uh, uw) for c, h, w, uh, uw in srcs]
    cin = sum(s_[0] for s_ in srcs)
    w = torch.randn((64, cin, 3, 3), generator=g) / np.sqrt(cin * 9.0)
    b = torch.randn((64,), generator=g) * 0.1
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda))
    Ho, Wo = srcs[0][1] * srcs[0][3] // stride[0], srcs[0][2] * srcs[0][4] // stride[1]
    res = ops.Blocked.from_nchw(torch.randn((N, 64, Ho, Wo), generator=g).to(cuda)) if res_mode else None
    kw = dict(relu=True, residual=res, res_mode=res_mode)
    kw.update({"out_dtype": torch.float32} if f32_out else {"out_blocked": True})
    raw = lambda o: o.view(torch.int32) if f32_out else o.t.view(torch.int16)
    try:
        _lib.lib().tpspp_conv_set_tuning(2)
        want = raw(ops.conv2d_bf16(xs, cw, stride, **kw)).clone()
    finally:
        _lib.lib().tpspp_conv_set_tuning(0)
    for _ in range(2):                                      # twice: no state survives a launch
        got = raw(ops.conv2d_bf16(xs, cw, stride, **kw))
        assert torch.equal(got, want)


WIDE_CASES = [
    # name, C (= Cin = Cout), H, W, res_mode, relu, N      (8x32: one image per workgroup; 4x16: four images per workgroup)
    ("layer3 128->128 @8x32 + residual", 128, 8, 32, 2, True, 37),
    ("layer4 256->256 @8x32 + residual", 256, 8, 32, 2, True, 9),
    ("layer5 512->512 @4x16 + residual, ragged group", 512, 4, 16, 2, True, 6),
    ("256->256 @8x32, no residual, no ReLU", 256, 8, 32, 0, False, 3),
    ("128->128 @4x16, residual after the ReLU, one image", 128, 4, 16, 1, True, 1),
    ("full machine 256->256 @8x32", 256, 8, 32, 2, True, 515),
    ("last block 512->512 @4x16 + residual, fp32 NCHW out", 512, 4, 16, 2, True, 7),
    ("256->256 @8x32, fp32 NCHW out, no residual", 256, 8, 32, 0, True, 2),
]


@pytest.mark.parametrize("name,C,H,W,res_mode,relu,N", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_conv3_wide_kernel_is_the_tiled_kernel_bit_for_bit(cuda, name, C, H, W, res_mode, relu, N):
    """tpspp_conv3_wide.hip (round 6: 128 x 64 wavefront tiles, the weight streamed from L2 into registers five taps ahead, the
    patch by LDS-DMA into a ring of three buffers) against the tiled kernel on the same blocked tensors
    (tpspp_conv_set_tuning bit 2 switches it off): the backbone's stage 3 - 5 shapes, with and without the blocked residual,
    a ragged last image group, more workgroups than the machine holds at once; twice (no state survives a launch); and
    against the CPU reference on bf16-rounded operands."""
    from tps_pp_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(len(name) + C)
    x = torch.randn((N, C, H, W), generator=g)
    w = torch.randn((C, C, 3, 3), generator=g) / np.sqrt(C * 9.0)
    b = torch.randn((C,), generator=g) * 0.1
    r = torch.randn((N, C, H, W), generator=g) if res_mode else None
    xs = [(ops.Blocked.from_nchw(x.to(cuda)), 1, 1)]
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda))
    res = ops.Blocked.from_nchw(r.to(cuda)) if res_mode else None
    f32_out = "fp32 NCHW" in name
    kw = dict(relu=relu, residual=res, res_mode=res_mode)
    kw.update({"out_dtype": torch.float32} if f32_out else {"out_blocked": True})
    raw = (lambda o: o.view(torch.int32)) if f32_out else (lambda o: o.t.view(torch.int16))
    try:
        _lib.lib().tpspp_conv_set_tuning(4)
        want = raw(ops.conv2d_bf16(xs, cw, (1, 1), **kw)).clone()
    finally:
        _lib.lib().tpspp_conv_set_tuning(0)
    for _ in range(2):
        got = ops.conv2d_bf16(xs, cw, (1, 1), **kw)
        assert torch.equal(raw(got), want), name
    if N <= 40:
        ref = ref_conv([(rb(x), 1, 1)], w, b, 3, (1, 1), relu, rb(r) if res_mode else None, res_mode)
        out = (got if f32_out else got.nchw()).float().cpu()
        assert float((out - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("C,H,N,relu", [(3, 32, 37, True), (3, 32, 771, True), (1, 8, 5, False), (3, 4, 1, True)])
def test_stem_kernel_is_the_tiled_kernel_bit_for_bit(cuda, C, H, N, relu):
    """tpspp_conv_stem.hip (round 6: the backbone's 3 -> 32 stem on the fp32 image, persistent workgroups, swapped matrix operands so
    that results leave as 16-byte pieces of the NCHW rows) against the tiled kernel (tpspp_conv_set_tuning bit 2 switches it off):
    the same bits; several trips per workgroup (771 images = 6168 tiles), a one-channel image, a single tile; and against the CPU
    reference on bf16-rounded operands."""
    from tps_pp_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(C * 100 + H + N)
    x = torch.randn((N, C, H, 128), generator=g)
    w = torch.randn((32, C, 3, 3), generator=g) / np.sqrt(C * 9.0)
    b = torch.randn((32,), generator=g) * 0.1
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda))
    xd = x.to(cuda)
    try:
        _lib.lib().tpspp_conv_set_tuning(4)
        want = ops.conv2d_bf16([xd], cw, 1, relu=relu).view(torch.int16).clone()
    finally:
        _lib.lib().tpspp_conv_set_tuning(0)
    for _ in range(2):
        got = ops.conv2d_bf16([xd], cw, 1, relu=relu)
        assert got.dtype == torch.bfloat16 and torch.equal(got.view(torch.int16), want)
    if N <= 40:
        ref = ref_conv([(x, 1, 1)], w, b, 3, (1, 1), relu, None, 0)
        assert float((got.float().cpu() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("name,H,W,stride,res_mode,f32_out,N", [
    ("layer1 32->32 @16x64 + residual", 16, 64, (1, 1), 2, False, 301),
    ("layer1 first 32->32 s2 @32x128 -> 16x64 + residual", 32, 128, (2, 2), 2, False, 67),
    ("32->32 @16x64, fp32 NCHW out", 16, 64, (1, 1), 0, True, 5),
    ("32->32 @8x32 (8x32 tiles)", 8, 32, (1, 1), 1, False, 515),
])
def test_conv_bf16_persistent_kernel_with_32_output_channels(cuda, name, H, W, stride, res_mode, f32_out, N):
    """Round 6: the persistent 3x3 kernel also takes Cout = 32 (the backbone's first stage: the arranged weight's 64-channel tile
    is zero-padded, the tensors have four channel groups): bit for bit the tiled kernel, blocked and fp32-NCHW output, blocked
    residual, stride 2."""
    from tps_pp_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(len(name))
    xs = [(ops.Blocked.from_nchw(torch.randn((N, 32, H, W), generator=g).to(cuda)), 1, 1)]
    w = torch.randn((32, 32, 3, 3), generator=g) / np.sqrt(32 * 9.0)
    b = torch.randn((32,), generator=g) * 0.1
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda))
    Ho, Wo = H // stride[0], W // stride[1]
    res = ops.Blocked.from_nchw(torch.randn((N, 32, Ho, Wo), generator=g).to(cuda)) if res_mode else None
    kw = dict(relu=True, residual=res, res_mode=res_mode)
    kw.update({"out_dtype": torch.float32} if f32_out else {"out_blocked": True})
    raw = lambda o: o.view(torch.int32) if f32_out else o.t.view(torch.int16)
    try:
        _lib.lib().tpspp_conv_set_tuning(2)
        want = raw(ops.conv2d_bf16(xs, cw, stride, **kw)).clone()
    finally:
        _lib.lib().tpspp_conv_set_tuning(0)
    for _ in range(2):
        got = ops.conv2d_bf16(xs, cw, stride, **kw)
        assert tuple((got if f32_out else got.t).shape[:2]) == ((N, 32) if f32_out else (N, 4))
        assert torch.equal(raw(got), want)


@pytest.mark.parametrize("fg_dtype", ["bf16", "f32"])
def test_front_bf16_fused_against_cpu_reference(cuda, fg_dtype):
    """tpspp_front_bf16_fwd (down0 / down1 / down2 / cat + Upsample + down_feat in one register-chained kernel)
    against PyTorch-CPU on bf16-rounded operands, feat_grid computed from the rounded feat0 / feat1 / feat2; odd
    batch, a width of three 32-pixel segments (partial last workgroup)."""
    from tps_pp_amd import TPS_PP
    import cases
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    N, H, W = 3, 6, 96
    o0 = rb(t(synth.dyadic((N, 32, H, W), "fb.o0", 1)))
    o1 = rb(t(synth.dyadic((N, 32, H, W), "fb.o1", 1)))
    x = rb(t(synth.dyadic((N, 64, H // 2, W // 2), "fb.x", 1)))

    def cm(mod, v):
        return F.relu(F.conv2d(v.double(), rb(mod.conv.weight.detach()).double(), mod.conv.bias.detach().double())).float()
    f0, f1, f2 = cm(m.down0, o0), cm(m.down1, o1), cm(m.down2, x)
    up = F.interpolate(rb(f2), scale_factor=2, mode="nearest")
    fg = cm(m.down_feat, torch.cat((rb(f0), rb(f1), up), 1))
    m.to(cuda)
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}[fg_dtype]
    fw = ops.FrontWeightsBf16(m)
    g0, g1, g2, gg = ops.front_bf16(o0.to(cuda).bfloat16(), o1.to(cuda).bfloat16(), x.to(cuda).bfloat16(), fw, dt)
    assert gg.dtype == dt and g0.dtype == torch.bfloat16
    for got, ref, name in ((g0, f0, "feat0"), (g1, f1, "feat1"), (g2, f2, "feat2")):
        got = got.float().cpu()
        err = (got - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-6).all()), (name, float(err.max()))
        assert float((got == rb(ref)).float().mean()) > 0.98, name
    # feat_grid sees the occasional one-ulp flip of its inputs
    got = gg.float().cpu()
    scale = float(fg.abs().max())
    assert float((got - fg).abs().max()) <= 2.0 ** -7 * scale
    assert float((got - fg).abs().mean()) <= (2e-3 if fg_dtype == "bf16" else 2e-4) * scale


@pytest.mark.parametrize("name,srcs,cout,k,stride,relu,res_mode,res_dt,N", CASES, ids=[c[0] for c in CASES])
def test_conv_bf16x3_matches_fp32_reference(cuda, name, srcs, cout, k, stride, relu, res_mode, res_dt, N):
    """split3 ("bf16x3": hi*hi + hi*lo + lo*hi on fp32 tensors): against float64 PyTorch-CPU on the UNROUNDED fp32
    operands, to 2e-5 of the tensor's scale (plain bf16 sits at 2.5e-3, the fp32 MFMA kernel at 3e-7)."""
    xs = [(t(synth.dyadic((N, c, h, w), f"{name}.x{i}", 1)), uh, uw) for i, (c, h, w, uh, uw, _) in enumerate(srcs)]
    cin = sum(s_[0] for s_ in srcs)
    w = t(synth.dyadic((cout, cin, k, k), name + ".w", 1, 1.0 / np.sqrt(cin * k * k)))
    b = t(synth.dyadic((cout,), name + ".b", 1, 0.1))

    def ref_fp(res):
        xi = [F.interpolate(x.double(), scale_factor=(uh, uw), mode="nearest") if (uh, uw) != (1, 1) else x.double()
              for x, uh, uw in xs]
        y = F.conv2d(torch.cat(xi, 1), w.double(), b.double(), stride=stride, padding=(k - 1) // 2)
        if res_mode == 2:
            y = y + res.double()
        if relu:
            y = F.relu(y)
        if res_mode == 1:
            y = y + res.double()
        return y.float()
    res = t(synth.dyadic(tuple(ref_fp(torch.zeros(1)).shape if not res_mode else ref_conv(xs, w, b, k, stride, relu, None, 0).shape),
                         name + ".r", 1)) if res_mode else None
    ref = ref_fp(res)
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda), x3=True)
    got = ops.conv2d_bf16([(x.to(cuda), uh, uw) for x, uh, uw in xs], cw, stride, relu,
                          None if res is None else res.to(cuda), res_mode, out_dtype=torch.float32).cpu()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), name


def test_front_bf16x3_fused_against_fp32_reference(cuda):
    """split3 of tpspp_front_bf16_fwd (fp32 tensors, three-term bf16 split, feat0 / feat1 / feat2 chained without an
    intermediate rounding) against float64 PyTorch-CPU on the unrounded operands: 2e-5 of each tensor's scale."""
    from tps_pp_amd import TPS_PP
    import cases
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    N, H, W = 3, 6, 96
    o0 = t(synth.dyadic((N, 32, H, W), "fx.o0", 1))
    o1 = t(synth.dyadic((N, 32, H, W), "fx.o1", 1))
    x = t(synth.dyadic((N, 64, H // 2, W // 2), "fx.x", 1))

    def cm(mod, v):
        return F.relu(F.conv2d(v.double(), mod.conv.weight.detach().double(), mod.conv.bias.detach().double()))
    f0, f1, f2 = cm(m.down0, o0), cm(m.down1, o1), cm(m.down2, x)
    fg = cm(m.down_feat, torch.cat((f0, f1, F.interpolate(f2, scale_factor=2, mode="nearest")), 1))
    m.to(cuda)
    fw = ops.FrontWeightsBf16(m, x3=True)
    got = ops.front_bf16(o0.to(cuda), o1.to(cuda), x.to(cuda), fw)
    for g_, r_, name in zip(got, (f0, f1, f2, fg), ("feat0", "feat1", "feat2", "feat_grid")):
        assert g_.dtype == torch.float32
        err = float((g_.cpu().double() - r_).abs().max())
        assert err <= 2e-5 * float(r_.abs().max()), (name, err)


def test_front_bf16_persistent_trips_match_single_image_runs(cuda):
    """The fused front runs persistent workgroups (one per CU, a wavefront walking every 1024th row segment): a batch
    large enough for several trips per wavefront, and an odd one, must give exactly what image-by-image calls give."""
    from tps_pp_amd import TPS_PP
    torch.manual_seed(5)
    m = TPS_PP().eval().to(cuda)
    fw = ops.FrontWeightsBf16(m)
    N, H, W = 21, 32, 128                                   # 21 * 32 * 4 = 2688 segments: 2-3 trips per wavefront
    o0 = torch.randn(N, 32, H, W, device=cuda).bfloat16()
    o1 = torch.randn(N, 32, H, W, device=cuda).bfloat16()
    x = torch.randn(N, 64, H // 2, W // 2, device=cuda).bfloat16()
    for dt in (torch.bfloat16, torch.float32):
        whole = ops.front_bf16(o0, o1, x, fw, dt)
        for n in (0, 7, 20):
            one = ops.front_bf16(o0[n:n + 1].contiguous(), o1[n:n + 1].contiguous(), x[n:n + 1].contiguous(), fw, dt)
            for a, b in zip(whole, one):
                assert torch.equal(a[n:n + 1], b)




@pytest.mark.parametrize("name,srcs,cout,k,stride,relu,res_mode,res_dt,N", BLK_CASES, ids=[c[0] for c in BLK_CASES])
def test_conv_bf16_blocked_layout_is_the_same_arithmetic(cuda, name, srcs, cout, k, stride, relu, res_mode, res_dt, N):
    """Layout code 2 (ops.Blocked: (N, C/8, H, W, 8)) for sources, residual and output: the same values, bit for bit, as
    the NCHW bf16 call -- the layout changes how a patch is staged and how results leave, not what is computed."""
    xs = [(t(synth.dyadic((N, c, h, w), f"{name}.x{i}", 1)).to(cuda).to(torch.bfloat16), uh, uw)
          for i, (c, h, w, uh, uw, _) in enumerate(srcs)]
    cin = sum(s_[0] for s_ in srcs)
    w = t(synth.dyadic((cout, cin, k, k), name + ".w", 1, 1.0 / np.sqrt(cin * k * k)))
    b = t(synth.dyadic((cout,), name + ".b", 1, 0.1))
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda))
    plain = ops.conv2d_bf16(xs, cw, stride, relu=relu)
    res = None
    if res_mode:
        res = t(synth.dyadic(tuple(plain.shape), name + ".r", 1)).to(cuda).to(torch.bfloat16)
    want = ops.conv2d_bf16(xs, cw, stride, relu=relu, residual=res, res_mode=res_mode)
    bsrc = [(ops.Blocked.from_nchw(x), uh, uw) for x, uh, uw in xs]
    bres = ops.Blocked.from_nchw(res) if res is not None else None
    # every combination of blocked / NCHW sources, residual and output (mixed sources: only the first one blocked)
    got = ops.conv2d_bf16(bsrc, cw, stride, relu=relu, residual=bres, res_mode=res_mode, out_blocked=True)
    assert isinstance(got, ops.Blocked) and got.shape == tuple(want.shape)
    assert torch.equal(got.nchw().view(torch.int16), want.view(torch.int16)), "blocked in / blocked out"
    got2 = ops.conv2d_bf16(bsrc, cw, stride, relu=relu, residual=res, res_mode=res_mode)
    assert torch.equal(got2.view(torch.int16), want.view(torch.int16)), "blocked in / NCHW out"
    mixed = [bsrc[0]] + xs[1:]
    got3 = ops.conv2d_bf16(mixed, cw, stride, relu=relu, residual=bres, res_mode=res_mode, out_blocked=True)
    assert torch.equal(got3.nchw().view(torch.int16), want.view(torch.int16)), "mixed sources"
    gotf = ops.conv2d_bf16(bsrc, cw, stride, relu=relu, residual=bres, res_mode=res_mode, out_dtype=torch.float32)
    wantf = ops.conv2d_bf16(xs, cw, stride, relu=relu, residual=res, res_mode=res_mode, out_dtype=torch.float32)
    assert torch.equal(gotf.view(torch.int32), wantf.view(torch.int32)), "blocked in / fp32 out"
    with pytest.raises(ValueError):
        ops.conv2d_bf16(bsrc, ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda), x3=True), stride)


@pytest.mark.parametrize("fg_dtype", [torch.bfloat16, torch.float32])
def test_front_bf16_blocked_outputs(cuda, fg_dtype):
    from tps_pp_amd import TPS_PP
    m = TPS_PP().to(cuda).eval()
    fw = ops.FrontWeightsBf16(m, False)
    g = torch.Generator(device=cuda).manual_seed(3)
    n = 5
    o0 = torch.rand((n, 32, 32, 128), generator=g, device=cuda).to(torch.bfloat16)
    o1 = torch.rand((n, 32, 32, 128), generator=g, device=cuda).to(torch.bfloat16)
    x = torch.rand((n, 64, 16, 64), generator=g, device=cuda).to(torch.bfloat16)
    a = ops.front_bf16(o0, o1, x, fw, fg_dtype)
    b = ops.front_bf16(o0, o1, x, fw, fg_dtype, blocked=True)
    for i in range(3):
        assert isinstance(b[i], ops.Blocked) and torch.equal(b[i].nchw().view(torch.int16), a[i].view(torch.int16)), i
    assert torch.equal(a[3], b[3])


def test_front_bf16x3_blocked_outputs(cuda):
    """The three-term-split front with feat0 / feat1 / feat2 in the fp32 blocked layout (16-byte stores of the result
    registers): the same bits as its NCHW outputs; feat_grid unchanged."""
    from tps_pp_amd import TPS_PP
    m = TPS_PP().to(cuda).eval()
    fw = ops.FrontWeightsBf16(m, True)
    g = torch.Generator(device=cuda).manual_seed(4)
    n = 5
    o0 = torch.rand((n, 32, 32, 128), generator=g, device=cuda)
    o1 = torch.rand((n, 32, 32, 128), generator=g, device=cuda)
    x = torch.rand((n, 64, 16, 64), generator=g, device=cuda)
    a = ops.front_bf16(o0, o1, x, fw)
    b = ops.front_bf16(o0, o1, x, fw, blocked=True)
    for i in range(3):
        assert isinstance(b[i], ops.Blocked32) and torch.equal(b[i].nchw().view(torch.int32), a[i].view(torch.int32)), i
    assert torch.equal(a[3], b[3])


@pytest.mark.parametrize("N,H", [(1, 32), (3, 32), (5, 8), (2, 2), (37, 32), (530, 32)])
def test_down_fused_is_front_then_stride2_conv_bit_for_bit(cuda, N, H):
    """tpspp_down_fused_bf16_fwd (down0 + down0_1 in one kernel, the 1x1 result only in LDS) against the two-kernel route:
    the front's blocked feat0 / feat1 followed by the 3x3 stride-2 convolution.  Whole images per workgroup (N = 530),
    strips of output rows with the row above recomputed (small N), a single output row (H = 2); signed inputs so that
    ReLU and the zero padding of the intermediate map (not of the input) both matter."""
    from tps_pp_amd import TPS_PP
    torch.manual_seed(11)
    m = TPS_PP().eval().to(cuda)
    with torch.no_grad():
        for c in (m.down0, m.down1, m.down0_1, m.down1_1):
            c.conv.bias.uniform_(-0.5, 0.5)                 # relu(b0) != 0: padding must be applied to feat, not to `in`
    fw = ops.FrontWeightsBf16(m)
    cw0 = ops.prep_conv_weight_bf16(m.down0_1.conv.weight, conv_bias=m.down0_1.conv.bias)
    cw1 = ops.prep_conv_weight_bf16(m.down1_1.conv.weight, conv_bias=m.down1_1.conv.bias)
    g = torch.Generator(device=cuda).manual_seed(N * 100 + H)
    o0 = torch.randn((N, 32, H, 128), generator=g, device=cuda).bfloat16()
    o1 = torch.randn((N, 32, H, 128), generator=g, device=cuda).bfloat16()
    x = torch.randn((N, 64, H // 2, 64), generator=g, device=cuda).bfloat16()
    f0, f1, f2, fg = ops.front_bf16(o0, o1, x, fw, torch.bfloat16, blocked=True)
    want0 = ops.conv2d_bf16([f0], cw0, 2, out_blocked=True)
    want1 = ops.conv2d_bf16([f1], cw1, 2, out_blocked=True)
    got0 = ops.down_fused_bf16(o0, fw.w0, fw.b0, cw0)
    got1 = ops.down_fused_bf16(o1, fw.w1, fw.b1, cw1)
    assert isinstance(got0, ops.Blocked) and got0.shape == want0.shape
    assert torch.equal(got0.t.view(torch.int16), want0.t.view(torch.int16))
    assert torch.equal(got1.t.view(torch.int16), want1.t.view(torch.int16))
    assert float(got0.t.float().abs().max()) > 0
    # the front without the feat0 / feat1 stores: same feat2 and feat_grid
    n0, n1, g2, gg = ops.front_bf16(o0, o1, x, fw, torch.bfloat16, blocked=True, store01=False)
    assert n0 is None and n1 is None
    assert torch.equal(g2.t.view(torch.int16), f2.t.view(torch.int16)) and torch.equal(gg.view(torch.int16), fg.view(torch.int16))
    gg32 = ops.front_bf16(o0, o1, x, fw, torch.float32, blocked=True, store01=False)[3]
    assert torch.equal(gg32, ops.front_bf16(o0, o1, x, fw, torch.float32, blocked=True)[3])


def test_front_bf16_nan_input_stays_in_its_pixel(cuda):
    """NaN inputs are outside the parity contract (include/tpspp.h, tpspp_front_bf16_fwd: the bf16 form's ReLU is a signed
    16-bit max after the rounding -- a NaN with the sign bit clear propagates as torch.relu's does, one with the sign bit
    set becomes +0).  What IS pinned: a NaN in one input pixel makes that pixel's feat0 channels NaN or +0 and nothing
    else -- every other element of all four outputs is bit for bit what the clean input gives (1x1 convolutions: no
    neighbour may see it), in the bf16 form and in the fused down kernel's 3x3 neighbourhood."""
    from tps_pp_amd import TPS_PP
    torch.manual_seed(13)
    m = TPS_PP().eval().to(cuda)
    fw = ops.FrontWeightsBf16(m)
    N, H, W = 2, 8, 128
    g = torch.Generator(device=cuda).manual_seed(5)
    o0 = torch.randn((N, 32, H, W), generator=g, device=cuda).bfloat16()
    o1 = torch.randn((N, 32, H, W), generator=g, device=cuda).bfloat16()
    x = torch.randn((N, 64, H // 2, W // 2), generator=g, device=cuda).bfloat16()
    clean = ops.front_bf16(o0, o1, x, fw, torch.bfloat16)
    for nan_bits in (0x7FC0, 0xFFC0):                       # quiet NaN, sign clear / set
        bad = o0.clone()
        bad.view(torch.int16)[1, 7, 3, 77] = nan_bits - (1 << 16) if nan_bits & 0x8000 else nan_bits
        got = ops.front_bf16(bad, o1, x, fw, torch.bfloat16)
        hit = torch.zeros((N, 1, H, W), dtype=torch.bool, device=cuda)
        hit[1, 0, 3, 77] = True
        f0 = got[0].float()
        at = f0[1, :, 3, 77]
        assert bool((torch.isnan(at) | (at == 0)).all())
        for k in (0, 3):                                    # feat0 and feat_grid: only the pixel itself may differ
            same = (got[k].view(torch.int16) == clean[k].view(torch.int16)) | hit
            assert bool(same.all()), (hex(nan_bits), k)
        assert torch.equal(got[1].view(torch.int16), clean[1].view(torch.int16))
        assert torch.equal(got[2].view(torch.int16), clean[2].view(torch.int16))
        # the fused down0 + down0_1 kernel: the 3x3 stride-2 neighbourhood of the pixel, nothing further
        cw0 = ops.prep_conv_weight_bf16(m.down0_1.conv.weight, conv_bias=m.down0_1.conv.bias)
        d_clean = ops.down_fused_bf16(o0, fw.w0, fw.b0, cw0).t
        d_bad = ops.down_fused_bf16(bad, fw.w0, fw.b0, cw0).t             # blocked (N, 8, H/2, W/2, 8)
        diff = (d_clean.view(torch.int16) != d_bad.view(torch.int16)).any(dim=4).any(dim=1)     # (N, H/2, W/2)
        ys, xs = torch.nonzero(diff[1], as_tuple=True)
        assert not bool(diff[0].any())
        assert all(abs(2 * int(y) - 3) <= 1 and abs(2 * int(xx) - 77) <= 1 for y, xx in zip(ys, xs)), (ys, xs)


def test_down_fused_argument_errors(cuda):
    from tps_pp_amd import TPS_PP
    m = TPS_PP().eval().to(cuda)
    fw = ops.FrontWeightsBf16(m)
    cw = ops.prep_conv_weight_bf16(m.down0_1.conv.weight, conv_bias=m.down0_1.conv.bias)
    with pytest.raises(ValueError):
        ops.down_fused_bf16(torch.zeros((1, 32, 32, 96), device=cuda).bfloat16(), fw.w0, fw.b0, cw)      # width
    with pytest.raises(ValueError):
        ops.down_fused_bf16(torch.zeros((1, 32, 31, 128), device=cuda).bfloat16(), fw.w0, fw.b0, cw)     # odd height
    with pytest.raises(ValueError):
        ops.down_fused_bf16(torch.zeros((1, 32, 32, 128), device=cuda), fw.w0, fw.b0, cw)                # fp32 map
    with pytest.raises(ValueError):
        ops.front_bf16(torch.zeros((1, 32, 32, 128), device=cuda).bfloat16(), torch.zeros((1, 32, 32, 128), device=cuda).bfloat16(),
                       torch.zeros((1, 64, 16, 64), device=cuda).bfloat16(), fw, store01=False)          # NCHW form
    assert ops.down_fused_bf16(torch.zeros((0, 32, 32, 128), device=cuda).bfloat16(), fw.w0, fw.b0, cw).shape == (0, 64, 16, 64)


@pytest.mark.parametrize("store01", [True, False])
def test_front_bf16_full_machine_matches_chunked_runs(cuda, store01):
    """The fused front prefetches the next segment's pieces under the current segment's matrix work and stores; since the
    round-4 fix it drains every request (`s_waitcnt vmcnt(0)`) before a segment's operands are used -- a counted wait does
    not skip younger stores safely.  With every CU busy (530 images) the result must still be what chunk-by-chunk calls
    give, run after run."""
    from tps_pp_amd import TPS_PP
    torch.manual_seed(9)
    m = TPS_PP().eval().to(cuda)
    fw = ops.FrontWeightsBf16(m)
    N = 530
    g = torch.Generator(device=cuda).manual_seed(77)
    o0 = torch.randn((N, 32, 32, 128), generator=g, device=cuda).bfloat16()
    o1 = torch.randn((N, 32, 32, 128), generator=g, device=cuda).bfloat16()
    x = torch.randn((N, 64, 16, 64), generator=g, device=cuda).bfloat16()
    for rep in range(3):
        whole = ops.front_bf16(o0, o1, x, fw, torch.bfloat16, blocked=True, store01=store01)
        for lo in range(0, N, 53):
            part = ops.front_bf16(o0[lo:lo + 53].contiguous(), o1[lo:lo + 53].contiguous(), x[lo:lo + 53].contiguous(), fw,
                                  torch.bfloat16, blocked=True, store01=store01)
            for a, b in zip(whole, part):
                if a is None:
                    continue
                a = a.t if isinstance(a, ops.Blocked) else a
                b = b.t if isinstance(b, ops.Blocked) else b
                assert torch.equal(a[lo:lo + 53].view(torch.int16), b.view(torch.int16)), (rep, lo)


C1X1_CASES = [
    # name, Cin, Cout, H, W, relu, N          (256-pixel tiles: 1 per 8x32 image, 4 per 16x64 image; 256 persistent workgroups)
    ("layer4 256->256 @8x32, 2-3 trips", 256, 256, 8, 32, True, 530),
    ("layer4 first 128->256 @8x32", 128, 256, 8, 32, True, 37),
    ("layer3 128->128 @8x32, no relu", 128, 128, 8, 32, False, 300),
    ("layer2 64->64 @16x64, ragged trips", 64, 64, 16, 64, True, 129),
    ("layer3 first 64->128 @16x64, one image", 64, 128, 16, 64, True, 1),
    # round 6: layers whose whole weight does not fit the LDS run as output-channel slices (2 / 4 launches); 4x16 maps (a tile
    # spans several images)
    ("layer5 first 256->512 @8x32, two slices", 256, 512, 8, 32, True, 261),
    ("128->256 @4x16: a tile spans several images", 128, 256, 4, 16, False, 36),
]


@pytest.mark.parametrize("name,cin,cout,H,W,relu,N", C1X1_CASES, ids=[c[0] for c in C1X1_CASES])
def test_conv1x1_blocked_kernel_is_the_tiled_kernel_bit_for_bit(cuda, name, cin, cout, H, W, relu, N):
    """tpspp_conv1x1_blk.hip (1x1 layers between blocked maps: every activation read once, the weight in LDS) against the
    tiled kernel on the same tensors (tpspp_conv_set_tuning bit 1 switches it off), and against float64 on bf16-rounded
    operands."""
    from tps_pp_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(cin + cout + N)
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 1, 1), generator=g) / np.sqrt(cin)
    b = torch.randn((cout,), generator=g) * 0.2
    xb = ops.Blocked.from_nchw(x.to(cuda))
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda))
    try:
        _lib.lib().tpspp_conv_set_tuning(2)
        want = ops.conv2d_bf16([xb], cw, 1, relu=relu, out_blocked=True).t.view(torch.int16).clone()
    finally:
        _lib.lib().tpspp_conv_set_tuning(0)
    for _ in range(2):
        got = ops.conv2d_bf16([xb], cw, 1, relu=relu, out_blocked=True)
        assert torch.equal(got.t.view(torch.int16), want)
    n = min(N, 3)
    ref = torch.einsum("oc,nchw->nohw", w.view(cout, cin).bfloat16().double(), x[:n].bfloat16().double()) + b.double().view(1, -1, 1, 1)
    if relu:
        ref = ref.clamp_min(0)
    err = (got.nchw()[:n].double().cpu() - ref).abs().max() / ref.abs().max()
    assert float(err) < 6e-3                                # one bf16 rounding of the result


@pytest.mark.parametrize("N,H", [(1, 32), (3, 32), (5, 8), (2, 2), (270, 32)])
def test_down_fused_x3_is_front_then_stride2_conv_bit_for_bit(cuda, N, H):
    """tpspp_down_fused_x3_fwd (the three-term split form: fp32 maps) against front_x3's blocked feat0 / feat1 followed by the
    three-term 3x3 stride-2 convolution; whole images per workgroup (N = 270: some workgroups take two), strips, one row."""
    from tps_pp_amd import TPS_PP
    torch.manual_seed(12)
    m = TPS_PP().eval().to(cuda)
    with torch.no_grad():
        for c in (m.down0, m.down1, m.down0_1, m.down1_1):
            c.conv.bias.uniform_(-0.5, 0.5)
    fw = ops.FrontWeightsBf16(m, True)
    cw0 = ops.prep_conv_weight_bf16(m.down0_1.conv.weight, conv_bias=m.down0_1.conv.bias, x3=True)
    cw1 = ops.prep_conv_weight_bf16(m.down1_1.conv.weight, conv_bias=m.down1_1.conv.bias, x3=True)
    g = torch.Generator(device=cuda).manual_seed(N * 100 + H + 1)
    o0 = torch.randn((N, 32, H, 128), generator=g, device=cuda)
    o1 = torch.randn((N, 32, H, 128), generator=g, device=cuda)
    x = torch.randn((N, 64, H // 2, 64), generator=g, device=cuda)
    f0, f1, f2, fg = ops.front_bf16(o0, o1, x, fw, blocked=True)
    want0 = ops.conv2d_bf16([f0], cw0, 2, out_dtype=torch.float32, out_blocked=True)
    want1 = ops.conv2d_bf16([f1], cw1, 2, out_dtype=torch.float32, out_blocked=True)
    got0 = ops.down_fused_bf16(o0, fw.w0, fw.b0, cw0)
    got1 = ops.down_fused_bf16(o1, fw.w1, fw.b1, cw1)
    assert isinstance(got0, ops.Blocked32) and got0.shape == want0.shape
    assert torch.equal(got0.t.view(torch.int32), want0.t.view(torch.int32))
    assert torch.equal(got1.t.view(torch.int32), want1.t.view(torch.int32))
    assert float(got0.t.abs().max()) > 0
    n0, n1, g2, gg = ops.front_bf16(o0, o1, x, fw, blocked=True, store01=False)
    assert n0 is None and n1 is None
    assert torch.equal(g2.t.view(torch.int32), f2.t.view(torch.int32)) and torch.equal(gg.view(torch.int32), fg.view(torch.int32))


def test_backbone_hands_tpspp_blocked_maps_with_the_same_bits(cuda):
    """Round 6: in the bf16 configuration our backbone keeps its stem output and its first stage's result in the blocked layout
    when `tpsnet` is this package's TPS_PP in the 'ResNet45' wiring (only its down convolutions read them): the stem kernel's
    blocked epilogue, the blocked 1x1 kernel at 32 input channels, the persistent 3x3 kernel at 32 output channels.  Same bits
    as the NCHW hand-over a foreign backbone would make -- feature map, rectified map, control-point score."""
    import tps_pp_amd as P
    torch.manual_seed(3)
    bb = P.build_backbone(dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3], strides=[2, 1, 2, 1, 2])).eval().to(cuda)
    tps = P.TPS_PP(variant="ResNet45").eval().to(cuda)
    bb.compute_dtype = torch.bfloat16
    img = torch.rand(5, 3, 32, 128, device=cuda) * 2 - 1
    seen = []
    orig_fwd = tps.forward

    def spy(x, outs, **kw):
        seen.append([type(o).__name__ for o in outs])
        return orig_fwd(x, outs, **kw)
    tps.forward = spy
    with torch.no_grad():
        got = bb(img, tpsnet=tps, test=True)
        tps.accepts_blocked_outs = lambda: False
        want = bb(img, tpsnet=tps, test=True)
    assert seen == [["Blocked", "Blocked"], ["Tensor", "Tensor"]], seen
    assert torch.equal(got["output"].view(torch.int32), want["output"].view(torch.int32))
    assert torch.equal(got["img_ref"].view(torch.int16), want["img_ref"].view(torch.int16))
    assert torch.isfinite(got["output"]).all()


def test_blocked_to_nchw_kernel(cuda):
    """`tpspp_blocked_to_nchw_bf16` (`ops.Blocked.nchw_hip`) = the PyTorch permutation, bit for bit; ragged sizes."""
    for n, c, h, w in ((3, 64, 16, 64), (1, 8, 8, 8), (5, 32, 4, 16), (2, 256, 8, 32)):
        g = torch.Generator(device="cpu").manual_seed(n + c)
        x = torch.randn((n, c, h, w), generator=g).to(cuda).bfloat16()
        b = ops.Blocked.from_nchw(x)
        got = b.nchw_hip()
        assert got.dtype == torch.bfloat16 and torch.equal(got.view(torch.int16), x.view(torch.int16))
        assert got._tpspp_blocked is b


@pytest.mark.parametrize("name,cin,cout,H,W,relu,f32_out,N", [
    ("layer5 first 256->512 @8x32", 256, 512, 8, 32, True, False, 37),
    ("layer4 256->256 @8x32, full machine", 256, 256, 8, 32, True, False, 515),
    ("layer5 512->512 @4x16, ragged group of eight", 512, 512, 4, 16, True, False, 530),
    ("512->512 @4x16, fp32 NCHW out, no relu, two images", 512, 512, 4, 16, False, True, 2),
])
def test_conv1x1_wide_kernel_is_the_tiled_kernel_bit_for_bit(cuda, name, cin, cout, H, W, relu, f32_out, N):
    """The wide-tile kernel's 1x1 form (tpspp_conv3_wide.hip, KS = 1: 32-channel chunks, two k-steps per chunk, the weight three
    k-steps ahead) for the 1x1 layers with >= 256 input channels, against the tiled kernel (tpspp_conv_set_tuning(6): neither the
    wide kernels nor the blocked 1x1 kernel): the same bits."""
    from tps_pp_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(cin + cout + N)
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 1, 1), generator=g) / np.sqrt(cin)
    b = torch.randn((cout,), generator=g) * 0.2
    xb = ops.Blocked.from_nchw(x.to(cuda))
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda))
    kw = {"out_dtype": torch.float32} if f32_out else {"out_blocked": True}
    raw = (lambda o: o.view(torch.int32)) if f32_out else (lambda o: o.t.view(torch.int16))
    try:
        _lib.lib().tpspp_conv_set_tuning(6)
        want = raw(ops.conv2d_bf16([xb], cw, 1, relu=relu, **kw)).clone()
    finally:
        _lib.lib().tpspp_conv_set_tuning(0)
    for _ in range(2):
        got = ops.conv2d_bf16([xb], cw, 1, relu=relu, **kw)
        assert torch.equal(raw(got), want)
