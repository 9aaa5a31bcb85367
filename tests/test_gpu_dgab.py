"""-m gpu: the fused DGAB kernels against the oracle's functional restatement of DGAB.py."""
import numpy as np
import pytest
import torch

import cases
from oracle import tpspp_oracle as TO
from tps_pp_amd import TPS_PP, ops

pytestmark = pytest.mark.gpu


def test_dgab_matches_oracle(cuda):
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    N = 3
    from tps_pp_amd import synth
    x = torch.from_numpy(synth.dyadic((N, 64, 16, 64), "dgab.x"))
    en = torch.from_numpy(synth.dyadic((N, 64, 2, 16), "dgab.en"))
    with torch.no_grad():
        ref = TO.dgab(dict(m.state_dict()), "TPE.atten.0", x, en.flatten(2).transpose(1, 2))
    m.to(cuda)
    dw = ops.DgabWeights(m.TPE.atten[0])
    got = ops.dgab(x.to(cuda), en.to(cuda).view(N, 64, 32), dw)
    err = (got.cpu() - ref).abs().max().item()
    assert err <= 1e-4, f"max abs err {err:.3e}"
    # and against the module's own PyTorch composition on the GPU
    with torch.no_grad():
        ref2 = m.TPE.atten[0](x.to(cuda), en.to(cuda).flatten(2).transpose(1, 2))
    assert (got - ref2).abs().max().item() <= 1e-4


def test_dgab_bf16_matches_bf16_oracle(cuda):
    """tpspp_dgab_bf16_fwd (proj / fc1 / fc2 on the bf16 matrix cores, everything else fp32) against the oracle
    that rounds the same operands to bfloat16; and, at bf16 resolution, against the fp32 oracle.  Odd batch."""
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    N = 5
    from tps_pp_amd import synth
    x = torch.from_numpy(synth.dyadic((N, 64, 16, 64), "dgab16.x"))
    en = torch.from_numpy(synth.dyadic((N, 64, 2, 16), "dgab16.en"))
    with torch.no_grad():
        ref16 = TO.dgab(dict(m.state_dict()), "TPE.atten.0", x, en.flatten(2).transpose(1, 2), bf16=True)
        ref32 = TO.dgab(dict(m.state_dict()), "TPE.atten.0", x, en.flatten(2).transpose(1, 2))
    m.to(cuda)
    dw = ops.DgabWeightsBf16(m.TPE.atten[0])
    got = ops.dgab_bf16(x.to(cuda), en.to(cuda).view(N, 64, 32), dw).cpu()
    scale = ref32.abs().max().item()
    e16, e32 = (got - ref16).abs(), (got - ref32).abs()
    # same roundings, different summation order / a flipped operand rounding here and there
    assert e16.max().item() <= 2e-3 * scale and e16.mean().item() <= 1e-4 * scale, (e16.max().item(), e16.mean().item())
    assert e32.max().item() <= 2e-2 * scale and e32.mean().item() <= 2e-3 * scale, (e32.max().item(), e32.mean().item())


def test_dgab_bf16x3_matches_fp32_oracle(cuda):
    """split3 of tpspp_dgab_bf16_fwd (three-term bf16 split on fp32 operands): the fp32 oracle's tolerance."""
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    N = 5
    from tps_pp_amd import synth
    x = torch.from_numpy(synth.dyadic((N, 64, 16, 64), "dgabx3.x"))
    en = torch.from_numpy(synth.dyadic((N, 64, 2, 16), "dgabx3.en"))
    with torch.no_grad():
        ref = TO.dgab(dict(m.state_dict()), "TPE.atten.0", x, en.flatten(2).transpose(1, 2))
    m.to(cuda)
    got = ops.dgab_bf16(x.to(cuda), en.to(cuda).view(N, 64, 32), ops.DgabWeightsBf16(m.TPE.atten[0], x3=True)).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 1e-4, f"max abs err {err:.3e}"


def test_score_matches_oracle(cuda):
    from tps_pp_amd import synth
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    N = 3
    de = torch.from_numpy(synth.dyadic((N, 64, 16, 64), "sc.de"))
    en = torch.from_numpy(synth.dyadic((N, 32, 64), "sc.en"))
    with torch.no_grad():
        ref = m.TPE.get_score(en, de)                      # PyTorch composition on the CPU
        p1 = m.TPE.p_linear(en)
    m.to(cuda)
    got = ops.score(de.to(cuda), p1.to(cuda).contiguous(), ops.ScoreWeights(m.TPE.feat_linear), m.TPE.scale)
    assert got.shape == ref.shape
    assert (got.cpu() - ref).abs().max().item() <= 2e-5
    # the three-term bf16 split of the same three products (tpspp_score_x3_fwd): the exact kernel's tolerance
    got3 = ops.score(de.to(cuda), p1.to(cuda).contiguous(), ops.ScoreWeights(m.TPE.feat_linear), m.TPE.scale, x3=True)
    assert (got3.cpu() - ref).abs().max().item() <= 2e-5
    # ragged pixel count (partial last workgroup)
    de2 = de[:, :, :5, :50].contiguous()
    with torch.no_grad():
        ref2 = m.cpu().TPE.get_score(en, de2)
    m.to(cuda)
    got2 = ops.score(de2.to(cuda), p1.to(cuda).contiguous(), ops.ScoreWeights(m.TPE.feat_linear), m.TPE.scale, x3=True)
    assert (got2.cpu() - ref2).abs().max().item() <= 2e-5


def test_cbam_and_point_stages_match_pytorch_cpu(cuda):
    from tps_pp_amd import synth
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    N = 5
    e3 = torch.from_numpy(np.abs(synth.dyadic((N, 64, 2, 16), "pt.e3")))
    with torch.no_grad():
        ref_cbam = m.MSFA.conv.atten(e3)
        en = e3.flatten(2).transpose(1, 2)
        ref_ctrl = m.TPE.localization_fc2(m.TPE.localization_fc1(en).view(N, -1)).view(N, 32, 2)
        ref_p = m.TPE.p_linear(en)
    m.to(cuda)
    got_cbam = ops.cbam(e3.to(cuda), m.MSFA.conv.atten)
    ctrl, p = ops.tpe_points(e3.to(cuda), m.TPE)
    assert (got_cbam.cpu() - ref_cbam).abs().max().item() <= 1e-6
    assert (ctrl.cpu() - ref_ctrl).abs().max().item() <= 2e-6
    assert (p.cpu() - ref_p).abs().max().item() <= 1e-5


def test_front_matches_pytorch_cpu(cuda):
    from tps_pp_amd import synth
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    N = 2
    o0 = torch.from_numpy(np.abs(synth.dyadic((N, 32, 32, 128), "fr.o0")))
    o1 = torch.from_numpy(np.abs(synth.dyadic((N, 32, 32, 128), "fr.o1")))
    x = torch.from_numpy(np.abs(synth.dyadic((N, 64, 16, 64), "fr.x")))
    with torch.no_grad():
        r0, r1, r2 = m.down0(o0), m.down1(o1), m.down2(x)
        rg = m.grid(r0, r1, r2)
    m.to(cuda)
    f0, f1, f2, fg = ops.front(o0.to(cuda), o1.to(cuda), x.to(cuda), ops.FrontWeights(m))
    for got, ref, nm in ((f0, r0, "feat0"), (f1, r1, "feat1"), (f2, r2, "feat2"), (fg, rg, "feat_grid")):
        assert got.shape == ref.shape
        assert (got.cpu() - ref).abs().max().item() <= 2e-5, nm


def test_point_stage_and_dgab_persistent_trips_match_small_batches(cuda):
    """tpe_points runs one persistent workgroup per CU with two images per trip, the DGAB chain eight wavefronts per
    workgroup walking 32-row tiles: a batch with several trips and an odd image count gives what slices of it give."""
    torch.manual_seed(6)
    m = TPS_PP().eval().to(cuda)
    N = 1031
    en = torch.randn(N, 64, 2, 16, device=cuda)
    ctrl, p = ops.tpe_points(en, m.TPE)
    for lo, hi in ((0, 3), (512, 517), (1028, 1031)):
        c1, p1 = ops.tpe_points(en[lo:hi].contiguous(), m.TPE)
        assert torch.equal(ctrl[lo:hi], c1) and torch.equal(p[lo:hi], p1)
    blk = m.TPE.atten[0]
    N2 = 37
    x = torch.randn(N2, 64, 16, 64, device=cuda)
    y = torch.randn(N2, 64, 32, device=cuda)
    for w, fn in ((ops.DgabWeights(blk), ops.dgab), (ops.DgabWeightsBf16(blk), ops.dgab_bf16),
                  (ops.DgabWeightsBf16(blk, x3=True), ops.dgab_bf16)):
        whole = fn(x, y, w)
        for lo, hi in ((0, 1), (17, 20), (36, 37)):
            assert torch.equal(whole[lo:hi], fn(x[lo:hi].contiguous(), y[lo:hi].contiguous(), w))
