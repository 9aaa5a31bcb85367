"""-m gpu: the fp32 MFMA convolution against PyTorch's CPU conv (the oracle's arithmetic) on the
shapes the TPS++ feature extractor, the localisation network and the backbone stem use."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tps_pp_amd import ops, synth

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def ref_conv(srcs, w, b, k, stride, relu, res, res_mode):
    xs = []
    for e in srcs:
        x, uh, uw = (e, 1, 1) if isinstance(e, torch.Tensor) else e
        xs.append(F.interpolate(x, scale_factor=(uh, uw), mode="nearest") if (uh, uw) != (1, 1) else x)
    y = F.conv2d(torch.cat(xs, 1), w, b, stride=stride, padding=(k - 1) // 2)
    if res_mode == 2:
        y = y + res
    if relu == 2:
        y = F.gelu(y)
    elif relu:
        y = F.relu(y)
    if res_mode == 1:
        y = y + res
    return y


CASES = [
    # name, sources [(C,H,W,uh,uw)], Cout, k, stride, relu, res_mode, N
    ("down0 1x1 32->64 @32x128", [(32, 32, 128, 1, 1)], 64, 1, (1, 1), True, 0, 3),
    ("down0_1 3x3 s2 64->64", [(64, 32, 128, 1, 1)], 64, 3, (2, 2), True, 0, 2),
    ("down_feat 1x1 cat(64,64,up 64)->64", [(64, 32, 128, 1, 1), (64, 32, 128, 1, 1), (64, 16, 64, 2, 2)], 64, 1, (1, 1), True, 0, 2),
    ("k_encoder.0 3x3 192->64 @16x64", [(64, 16, 64, 1, 1), (64, 16, 64, 1, 1), (64, 16, 64, 1, 1)], 64, 3, (1, 1), True, 0, 2),
    ("k_encoder.3 3x3 s(2,1) @4x16", [(64, 4, 16, 1, 1)], 64, 3, (2, 1), True, 0, 3),
    ("k_decoder.0 up(2,1)+3x3 + skip", [(64, 2, 16, 2, 1)], 64, 3, (1, 1), True, 1, 3),
    ("k_decoder.2 up2+3x3 + skip @16x64", [(64, 8, 32, 2, 2)], 64, 3, (1, 1), True, 1, 2),
    ("stem 3x3 3->32 @32x128 (folded BN)", [(3, 32, 128, 1, 1)], 32, 3, (1, 1), True, 0, 2),
    ("BasicBlock conv2 3x3 s2 + downsample residual", [(64, 32, 128, 1, 1)], 64, 3, (2, 2), True, 2, 2),
    ("localisation conv 3x3 3->64 @32x100 no bias", [(3, 32, 100, 1, 1)], 64, 3, (1, 1), False, 0, 2),
    ("wide 1x1 64->256 odd pixels", [(64, 7, 9, 1, 1)], 256, 1, (1, 1), False, 0, 2),
    ("Cout not multiple of 64", [(16, 9, 13, 1, 1)], 40, 3, (1, 1), True, 0, 1),
    # 4x16 outputs (last backbone stage): 128-pixel tiles that span two images, odd batch
    ("two-image tile 3x3 64->64 @4x16", [(64, 4, 16, 1, 1)], 64, 3, (1, 1), True, 2, 3),
    ("two-image tile 3x3 s2 32->64 @8x32 -> 4x16", [(32, 8, 32, 1, 1)], 64, 3, (2, 2), True, 0, 5),
    ("two-image tile 1x1 64->128 @4x16", [(64, 4, 16, 1, 1)], 128, 1, (1, 1), False, 1, 3),
    ("two-image tile 3x3 @2x16 two sources", [(8, 2, 16, 1, 1), (8, 1, 16, 2, 1)], 64, 3, (1, 1), True, 0, 4),
    # large 1x1 GEMMs -> 128x128 tiles
    ("wide 1x1 512->1536 one row of 8192", [(512, 1, 8192, 1, 1)], 1536, 1, (1, 1), False, 0, 1),
    ("wide 1x1 256->512 gelu+res @16x64 x 8", [(256, 16, 64, 1, 1)], 512, 1, (1, 1), 2, 1, 8),
    ("wide 1x1 K=40 Cout=132 ragged", [(40, 12, 44, 1, 1)], 132, 1, (1, 1), True, 2, 9),
    # few output pixels -> split-K skinny kernel (decoder steps, batches of feature vectors)
    ("skinny 1x1 512->512 one row of 512", [(512, 1, 512, 1, 1)], 512, 1, (1, 1), False, 1, 1),
    ("skinny 1x1 512->92 ragged", [(512, 1, 77, 1, 1)], 92, 1, (1, 1), True, 0, 1),
    ("skinny 1x1 K=100 odd sizes", [(100, 3, 5, 1, 1)], 37, 1, (1, 1), True, 2, 3),
    ("skinny 1x1 K=6", [(6, 1, 40, 1, 1)], 8, 1, (1, 1), False, 0, 2),
    # at most 32 output channels: the one-accumulator form of the tiled kernel (round 6)
    ("stem 3x3 3->32 @32x128", [(3, 32, 128, 1, 1)], 32, 3, (1, 1), True, 0, 2),
    ("BasicBlock 3x3 32->32 + residual @16x64", [(32, 16, 64, 1, 1)], 32, 3, (1, 1), True, 2, 3),
    ("3x3 s2 32->32 + residual, Cout=24 ragged", [(32, 16, 64, 1, 1)], 24, 3, (2, 2), True, 2, 3),
    ("1x1 32->32 @32x128", [(32, 32, 128, 1, 1)], 32, 1, (1, 1), True, 0, 2),
    # stride-2 3x3 tiles with upsampled / concatenated sources: the lean index bookkeeping's other paths (round 6)
    ("3x3 s2 on an up2 source @8x32 -> 16x64 -> 8x32", [(64, 8, 32, 2, 2)], 64, 3, (2, 2), True, 0, 3),
    ("3x3 s2 on cat(full, up(3,1)) @12x20", [(8, 12, 20, 1, 1), (8, 4, 20, 3, 1)], 40, 3, (2, 2), True, 2, 2),
    # strided 1x1 (the backbone's downsample branches; round 6: the tiled kernel staging only the pixels it uses)
    ("downsample 1x1 s2 256->512 @8x32 -> 4x16 (two-image tile, odd batch)", [(256, 8, 32, 1, 1)], 512, 1, (2, 2), False, 0, 5),
    ("downsample 1x1 s2 64->128 @16x64 -> 8x32", [(64, 16, 64, 1, 1)], 128, 1, (2, 2), False, 0, 3),
    ("downsample 1x1 s2 32->32 @32x128 -> 16x64", [(32, 32, 128, 1, 1)], 32, 1, (2, 2), True, 0, 2),
    ("downsample 1x1 s2 K=40 odd map 7x21 -> 4x11, Cout=36", [(40, 7, 21, 1, 1)], 36, 1, (2, 2), True, 2, 3),
]


@pytest.mark.parametrize("name,srcs,cout,k,stride,relu,res_mode,N", CASES, ids=[c[0] for c in CASES])
def test_conv_matches_cpu_reference(cuda, name, srcs, cout, k, stride, relu, res_mode, N):
    xs = [(t(synth.dyadic((N, c, h, w), f"{name}.x{i}", 1)), uh, uw) for i, (c, h, w, uh, uw) in enumerate(srcs)]
    cin = sum(s_[0] for s_ in srcs)
    w = t(synth.dyadic((cout, cin, k, k), name + ".w", 1, 1.0 / np.sqrt(cin * k * k)))
    b = None if "no bias" in name else t(synth.dyadic((cout,), name + ".b", 1, 0.1))
    y0 = ref_conv(xs, w, b, k, stride, relu, None, 0)
    res = t(synth.dyadic(tuple(y0.shape), name + ".r", 1)) if res_mode else None
    ref = ref_conv(xs, w, b, k, stride, relu, res, res_mode)
    cw = ops.prep_conv_weight(w.to(cuda), conv_bias=None if b is None else b.to(cuda),
                              src_channels=[s_[0] for s_ in srcs])
    from tps_pp_amd import _lib
    try:
        for force_generic in (0, 1):          # tiled kernel (where a tile fits) and generic kernel
            _lib.lib().tpspp_conv_set_tuning(force_generic)
            got = ops.conv2d([(x.to(cuda), uh, uw) for x, uh, uw in xs], cw, stride, relu,
                             None if res is None else res.to(cuda), res_mode)
            assert got.shape == ref.shape
            err = (got.cpu() - ref).abs().max().item()
            assert err <= 2e-5, f"{name} (generic={force_generic}): max abs err {err:.3e}"
    finally:
        _lib.lib().tpspp_conv_set_tuning(0)


def test_folded_batchnorm(cuda):
    N, cin, cout = 2, 32, 64
    x = t(synth.dyadic((N, cin, 16, 64), "bn.x"))
    w = t(synth.dyadic((cout, cin, 3, 3), "bn.w", 0, 1.0 / np.sqrt(cin * 9)))
    gamma = t(synth.dyadic((cout,), "bn.g", 0, 0.25, 1.0)); beta = t(synth.dyadic((cout,), "bn.b", 0, 0.1))
    mean = t(synth.dyadic((cout,), "bn.m", 0, 0.1)); var = t(synth.dyadic((cout,), "bn.v", 0, 0.25, 1.0))
    ref = F.relu(F.batch_norm(F.conv2d(x, w, None, padding=1), mean, var, gamma, beta, False, 0.1, 1e-5))
    cw = ops.prep_conv_weight(w.to(cuda), bn=tuple(v.to(cuda) for v in (gamma, beta, mean, var)))
    got = ops.conv2d([x.to(cuda)], cw, 1, True)
    assert (got.cpu() - ref).abs().max().item() <= 3e-5


@pytest.mark.parametrize("N,H", [(1, 32), (3, 32), (5, 8), (2, 2), (270, 32)])
def test_down_fused_f32_is_front_then_stride2_conv_bit_for_bit(cuda, N, H):
    """tpspp_down_fused_f32_fwd (exact fp32: down0 + down0_1 in one kernel, the 1x1 result only in LDS) against tpspp_front_fwd's
    feat0 / feat1 followed by the 3x3 stride-2 tpspp_conv2d_fwd; whole images per workgroup (N = 270: some workgroups take
    two), strips with a recomputed halo row, a single output row; signed inputs and non-zero biases (the padding applies to
    the intermediate map)."""
    from tps_pp_amd import TPS_PP
    torch.manual_seed(13)
    m = TPS_PP().eval().to(cuda)
    with torch.no_grad():
        for c in (m.down0, m.down1, m.down0_1, m.down1_1):
            c.conv.bias.uniform_(-0.5, 0.5)
    fw = ops.FrontWeights(m)
    cw0 = ops.prep_conv_weight(m.down0_1.conv.weight, conv_bias=m.down0_1.conv.bias)
    cw1 = ops.prep_conv_weight(m.down1_1.conv.weight, conv_bias=m.down1_1.conv.bias)
    g = torch.Generator(device=cuda).manual_seed(N * 100 + H + 2)
    o0 = torch.randn((N, 32, H, 128), generator=g, device=cuda)
    o1 = torch.randn((N, 32, H, 128), generator=g, device=cuda)
    x = torch.randn((N, 64, H // 2, 64), generator=g, device=cuda)
    f0, f1, f2, fg = ops.front(o0, o1, x, fw)
    want0 = ops.conv2d([f0], cw0, 2)
    want1 = ops.conv2d([f1], cw1, 2)
    got0 = ops.down_fused_f32(o0, fw.w0, fw.b0, cw0)
    got1 = ops.down_fused_f32(o1, fw.w1, fw.b1, cw1)
    assert got0.shape == want0.shape
    assert torch.equal(got0.view(torch.int32), want0.view(torch.int32))
    assert torch.equal(got1.view(torch.int32), want1.view(torch.int32))
    assert float(got0.abs().max()) > 0
    n0, n1, g2, gg = ops.front(o0, o1, x, fw, store01=False)
    assert n0 is None and n1 is None and torch.equal(g2, f2) and torch.equal(gg, fg)
