"""-m gpu: the module-level drop-ins on a real MI355X against the reference's golden outputs.

Two bars per module:
  * the transformation stage (hand-written HIP), fed the reference's own control points / score:
    BIT-EXACT against the reference's grid and warped tensors;
  * the whole forward (regressor still on PyTorch-ROCm library kernels, whose summation order is
    not the CPU's): warped tensors within the north-star tolerance 1e-4 on image-like inputs.
"""
import numpy as np
import pytest
import torch

import cases
from tps_pp_amd import TPS_PP, TPSPreprocessor, build_backbone

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_biteq(a, b, what):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    ne = bits(a) != bits(b)
    assert a.shape == b.shape and not ne.any(), \
        f"{what}: {int(ne.sum())} of {ne.size} differ, max abs {np.abs(a - b).max():.3e}"


def test_classic_module_against_reference(cuda):
    G = cases.load("classic_module")
    m = TPSPreprocessor(num_fiducial=cases.CL_F, img_size=cases.CL_HW,
                        rectified_img_size=cases.CL_HW, num_img_channel=3).eval()
    sd = cases.synth_state(m.state_dict(), 1, cases.g1_state_rule, cases.G1_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.to(cuda)
    img = dev(cases.g1_inputs()["img"], cuda)
    with torch.no_grad():
        out, grid, idx = m.rectify(img, dev(G["ctrl"], cuda), want_grid=True, want_idx=True)
        assert_biteq(grid, G["grid"], "grid from the reference's control points")
        assert_biteq(out, G["out"], "rectified image from the reference's control points")
        full = m(img)
        ctrl = m.LocalizationNetwork(img)
    assert np.abs(ctrl.cpu().numpy() - G["ctrl"]).max() < 1e-5
    assert np.abs(full.cpu().numpy() - G["out"]).max() <= TOL
    # the reference's own test (test_ocr_preprocessor.py:19-29): shape of a 1x1x32x100 forward
    p1 = TPSPreprocessor(20, (32, 100), (32, 100), 1).eval().to(cuda)
    with torch.no_grad():
        assert p1(torch.randn(1, 1, 32, 100, device=cuda)).shape == torch.Size([1, 1, 32, 100])


@pytest.mark.parametrize("variant,fname", [("ResNet45v2", "tpspp_module_v2"),
                                           ("ResNet45", "tpspp_module_v1")])
def test_tpspp_module_against_reference(cuda, variant, fname):
    G = cases.load(fname)
    m = build_backbone(dict(type="TPS_PP", variant=variant)).eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.to(cuda)
    inp = cases.g4_inputs(variant)
    x, outs = dev(inp["x"], cuda), [dev(o, cuda) for o in inp["outs"]]
    with torch.no_grad():
        ctrl, score, feat_grid = m.regress(x, outs)
        # transformation stage from the reference's control points and score
        out0, out1, grid = m.rectify(feat_grid.contiguous(), x, dev(G["ctrl"], cuda),
                                     dev(G["pc_score"], cuda), want_grid=True)
        res = m(x, outs)
    assert_biteq(grid, G["grid"], "grid")
    assert_biteq(out1, G["mp_img"], "mp_img (samples the raw input)")
    if variant == "ResNet45":
        assert_biteq(out0, G["output"], "output (feat_grid is the raw input in this wiring)")
    else:
        assert np.abs(out0.cpu().numpy() - G["output"]).max() <= TOL
    assert set(res) == {"output", "logits", "mp_img", "pc_score"} and res["logits"] is None
    assert np.abs(ctrl.cpu().numpy() - G["ctrl"]).max() < 2e-5
    assert np.abs(score.cpu().numpy() - G["pc_score"]).max() < 1e-4
    # whole forward, regressor on the GPU (exact-fp32 kernels): image-like inputs, the north-star's 1e-4
    e_mp = np.abs(res["mp_img"].cpu().numpy() - G["mp_img"]).max()
    e_out = np.abs(res["output"].cpu().numpy() - G["output"]).max()
    assert e_mp <= TOL and e_out <= TOL, (e_mp, e_out)
    assert res["output"].shape == torch.Size([cases.G4_N, 64, 16, 64])


STAGE_ORDER = ["feat_cat", "enc0", "enc1", "enc2", "enc3", "cbam", "dec0_sum", "dec1_sum", "dec2_sum", "dec3", "dgab"]


def _stage_errors(st, G):
    """max |HIP stage - reference intermediate| per stage, in the order of the forward pass.  The goldens are the
    reference's own forward-hook outputs (tests/golden/make_golden.py:140-180; `*_sub` = every 8th channel);
    k_decoder.i's hook fires before the skip addition (`tps_pp.py:165-167`), which the HIP convolution has fused in, so
    `dec{i}_sum` is compared with golden conv output + golden skip map (the fp32 addition the reference performs)."""
    sub = cases.sub
    want = {"feat_cat": G["feat_cat_sub"], "cbam": G["cbam"], "dgab": G["dgab_sub"], "dec3": G["dec3_conv_sub"]}
    for i in range(4):
        want[f"enc{i}"] = G[f"enc{i}_sub"]
    for i in range(3):
        want[f"dec{i}_sum"] = G[f"dec{i}_conv_sub"] + G[f"enc{2 - i}_sub"]
    errs = {}
    for name in STAGE_ORDER:
        got = st[name].cpu().numpy()
        got = got if name == "cbam" else sub(got)
        assert got.shape == want[name].shape, (name, got.shape, want[name].shape)
        errs[name] = float(np.abs(got - want[name]).max())
    return errs


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("bf16x3", 1e-4)])
def test_tpspp_stages_against_reference_intermediates(cuda, mode, tol):
    """Every stage of the regressor ON THE GPU against the reference's own intermediates (golden G4: forward hooks on
    `tps_pp.py:156-169` k_encoder / atten / k_decoder and `DGAB.py:58-77`), not only end to end: the exact-fp32 kernels
    within 2e-5, the three-term split within 1e-4.  Then one weight of one layer is perturbed: every stage before that layer
    still passes and the layer's own stage is the first to fail."""
    G = cases.load("tpspp_module_v2")
    m = build_backbone(dict(type="TPS_PP", variant="ResNet45v2")).eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.to(cuda)
    if mode == "bf16x3":
        m.compute_dtype = "bf16x3"
    inp = cases.g4_inputs("ResNet45v2")
    x, outs = dev(inp["x"], cuda), [dev(o, cuda) for o in inp["outs"]]
    with torch.no_grad():
        ctrl, score, _, st = m.regress_stages(x, outs)
    assert set(st) == set(STAGE_ORDER)
    errs = _stage_errors(st, G)
    assert all(e <= tol for e in errs.values()), errs
    assert np.abs(ctrl.cpu().numpy() - G["ctrl"]).max() < 2e-5
    # the hook leaves no state behind: a plain regress() afterwards records nothing
    with torch.no_grad():
        m.regress(x, outs)
    assert m._stage_tap is None
    # one perturbed weight per probe: the stage of that layer is the first that fails
    for layer, stage in ((m.MSFA.conv.k_encoder[2].conv, "enc2"), (m.MSFA.conv.k_decoder[1][1].conv, "dec1_sum"),
                         (m.down1_1.conv, "feat_cat")):
        with torch.no_grad():
            w = layer.weight
            old = w[8, :, 1, 1].clone()      # output channel 8 (the goldens keep every 8th channel), centre tap of EVERY
            w[8, :, 1, 1] = old + 0.25       # input channel (single input channels are dead behind their ReLU)
            _, _, _, st2 = m.regress_stages(x, outs)
            w[8, :, 1, 1] = old
        e2 = _stage_errors(st2, G)
        failing = [n for n in STAGE_ORDER if e2[n] > tol]
        assert failing and failing[0] == stage, (stage, e2)
    with torch.no_grad():
        _, _, _, st3 = m.regress_stages(x, outs)
    assert all(e <= tol for e in _stage_errors(st3, G).values())


@pytest.mark.parametrize("variant,fname", [("ResNet45v2", "tpspp_module_v2"), ("ResNet45", "tpspp_module_v1")])
def test_tpspp_module_bf16x3_meets_the_fp32_bar(cuda, variant, fname):
    """`compute_dtype = "bf16x3"` (fp32 tensors, three-term bf16 split in the convolutions) against the reference's
    outputs with the SAME tolerances as the exact-fp32 path (test_tpspp_module_against_reference)."""
    G = cases.load(fname)
    m = build_backbone(dict(type="TPS_PP", variant=variant)).eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.to(cuda)
    m.compute_dtype = "bf16x3"
    inp = cases.g4_inputs(variant)
    x, outs = dev(inp["x"], cuda), [dev(o, cuda) for o in inp["outs"]]
    with torch.no_grad():
        ctrl, score, _ = m.regress(x, outs)
        res = m(x, outs)
    assert res["output"].dtype == torch.float32
    assert np.abs(ctrl.cpu().numpy() - G["ctrl"]).max() < 2e-5
    assert np.abs(score.cpu().numpy() - G["pc_score"]).max() < 1e-4
    assert np.abs(res["mp_img"].cpu().numpy() - G["mp_img"]).max() <= TOL
    assert np.abs(res["output"].cpu().numpy() - G["output"]).max() <= TOL


def test_classic_module_bf16_localisation(cuda):
    """TPSPreprocessor with the localisation network's convolutions on the bf16 matrix cores
    (`LocalizationNetwork.compute_dtype`): control points and the rectified image against the reference's fp32 run
    (golden G1) at bf16 resolution -- the warp itself stays the fp32 kernel."""
    G = cases.load("classic_module")
    m = TPSPreprocessor(num_fiducial=cases.CL_F, img_size=cases.CL_HW,
                        rectified_img_size=cases.CL_HW, num_img_channel=3).eval()
    sd = cases.synth_state(m.state_dict(), 1, cases.g1_state_rule, cases.G1_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.to(cuda)
    m.LocalizationNetwork.compute_dtype = torch.bfloat16
    img = dev(cases.g1_inputs()["img"], cuda)
    with torch.no_grad():
        ctrl = m.LocalizationNetwork(img)
        out = m(img)
    assert np.abs(ctrl.cpu().numpy() - G["ctrl"]).max() <= 5e-3
    assert np.abs(out.cpu().numpy() - G["out"]).mean() <= 5e-3 * np.abs(G["out"]).max()
    # "bf16x3": the three-term split keeps the exact path's tolerances (test_classic_module_against_reference)
    m.LocalizationNetwork.compute_dtype = "bf16x3"
    with torch.no_grad():
        ctrl = m.LocalizationNetwork(img)
        out = m(img)
    assert np.abs(ctrl.cpu().numpy() - G["ctrl"]).max() < 1e-5
    assert np.abs(out.cpu().numpy() - G["out"]).max() <= TOL


@pytest.mark.parametrize("variant,fname", [("ResNet45v2", "tpspp_module_v2"), ("ResNet45", "tpspp_module_v1")])
def test_tpspp_module_bf16(cuda, variant, fname):
    """The bf16 configuration (bf16 inputs -> bf16 MFMA convolutions, fp32 from the control points on) against
    the reference's fp32 outputs (golden G4 / G5) at bf16 resolution, fp32 inputs with `compute_dtype` set, and
    the refusal of a module cast to bfloat16."""
    G = cases.load(fname)
    m = build_backbone(dict(type="TPS_PP", variant=variant)).eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.to(cuda)
    inp = cases.g4_inputs(variant)
    xb, outsb = dev(inp["x"], cuda).to(torch.bfloat16), [dev(o, cuda).to(torch.bfloat16) for o in inp["outs"]]
    with torch.no_grad():
        ctrl, score, _ = m.regress(xb, outsb)
        res = m(xb, outsb)
    assert res["output"].dtype == torch.bfloat16 and res["mp_img"].dtype == torch.bfloat16
    assert ctrl.dtype == torch.float32 and np.abs(ctrl.cpu().numpy() - G["ctrl"]).max() < 1e-4
    assert np.abs(score.float().cpu().numpy() - G["pc_score"]).max() < 2e-3
    for k in ("output", "mp_img"):       # inputs rounded to bf16 (2^-9 relative) + outputs rounded to bf16
        ref = G[k]
        assert np.abs(res[k].float().cpu().numpy() - ref).max() <= 2.0 ** -6 * np.abs(ref).max(), k
    # fp32 tensors, bf16 convolutions: fp32 outputs
    m.compute_dtype = torch.bfloat16
    with torch.no_grad():
        res32 = m(dev(inp["x"], cuda), [dev(o, cuda) for o in inp["outs"]])
    assert res32["output"].dtype == torch.float32
    assert np.abs(res32["output"].cpu().numpy() - G["output"]).max() <= 2.0 ** -7 * np.abs(G["output"]).max()
    m.compute_dtype = None
    with pytest.raises(TypeError):
        m.to(torch.bfloat16)(xb, outsb)


def test_backbone_stem_and_tps_call_site(cuda):
    """Stem + layer1 + layer2 of ResNetABI_v2_large on the MFMA conv kernels (BatchNorm folded) against
    the reference's outputs, and the `tpsnet(x, outs)` call contract (resnet_v2_large.py:183-191)."""
    from tps_pp_amd import ResNetABI_v2_large
    G = cases.load("backbone_stem")
    m = ResNetABI_v2_large(strides=cases.G7_STRIDES).eval()
    keep = {k: v for k, v in m.state_dict().items() if k.startswith(("conv1.", "bn1.", "layer1.", "layer2."))}
    sd = cases.synth_state(keep, 7, cases.backbone_state_rule)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.to(cuda)
    got = {}

    class Spy(torch.nn.Module):
        def forward(self, x, outs, **kw):
            got["x"], got["outs"] = x, list(outs)
            return {"output": x}
    with torch.no_grad():
        res = m(dev(cases.g7_inputs()["img"], cuda), Spy())
    assert set(res) == {"output", "img_ref"} and res["output"].shape == torch.Size([cases.G7_N, 512, 4, 16])
    # 1e-4 of each map's scale (max |reference value|, never below 1): the stage outputs of 3 + 4 residual blocks with
    # hash-generated weights are not O(1) like an image
    for name, a, b in (("x", got["x"].cpu().numpy(), G["x"]),
                       ("outs[0]", got["outs"][0].cpu().numpy()[:, ::4], G["outs0_sub"]),
                       ("outs[1]", got["outs"][1].cpu().numpy()[:, ::4], G["outs1_sub"])):
        scale = max(1.0, float(np.abs(b).max()))
        err = float(np.abs(a - b).max())
        assert err <= 1e-4 * scale, f"{name}: max |err| {err:.3e} against scale {scale:.3f}"
    # end to end with the real rectifier in the loop
    tps = build_backbone(dict(type="TPS_PP")).eval().to(cuda)
    with torch.no_grad():
        res = m(dev(cases.g7_inputs()["img"], cuda), tps)
    assert res["img_ref"].shape == torch.Size([cases.G7_N, 64, 16, 64]) and torch.isfinite(res["output"]).all()


def test_nrtr_modality_transform_against_reference(cuda):
    from tps_pp_amd import NRTRModalityTransform
    G = cases.load("nrtr_stem")
    m = NRTRModalityTransform().eval()
    sd = cases.synth_state(m.state_dict(), 8, cases.nrtr_state_rule)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    img = cases.g8_inputs()["img"]
    with torch.no_grad():
        cpu = m._forward_torch(torch.from_numpy(img))     # PyTorch composition on the CPU (test hook)
        got = m.to(cuda)(dev(img, cuda))
    assert np.abs(cpu.numpy() - G["out"]).max() <= 1e-4
    assert got.shape == torch.Size([cases.G8_N, 512, 1, 25])
    assert np.abs(got.cpu().numpy() - G["out"]).max() <= 1e-4
