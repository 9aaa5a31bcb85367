"""-m gpu: backward of the fused warp against the reference's own autograd
(golden G14) and the CPU oracle.  Gradients are sums over many taps whose fp32 accumulation order is free (the
reference's own order is its CPU kernel's), so the bar is a relative tolerance, stated per tensor.  At the TPS_PP
geometry the input gradients are accumulated in 64-bit fixed point (exact, order-independent): those are also
checked for run-to-run bit equality and against float64 at a tolerance no fp32 accumulation would meet."""
import numpy as np
import pytest
import torch

import cases
from oracle import tps_oracle as O
from tps_pp_amd import TPS_PP, TPSPreprocessor, ops

pytestmark = pytest.mark.gpu


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def close(got, want, rtol, what):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert got.shape == want.shape and err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def test_classic_backward_against_reference(cuda):
    G = cases.load("warp_backward")
    inp, gi = cases.g2_inputs(), cases.g14_inputs()
    c = O.classic_constants(cases.CL_F, cases.CL_HW)
    inv, ph = dev(c["inv_delta_C"], cuda), dev(c["P_hat"], cuda)
    img = dev(inp["img_smooth"], cuda).requires_grad_(True)
    ctrl = dev(inp["ctrl"], cuda).requires_grad_(True)
    out = ops.warp_autograd(img, ctrl, inv, ph, cases.CL_HW)
    (out * dev(gi["g_out_cl"], cuda)).sum().backward()
    close(img.grad, G["cl_g_img"], 2e-5, "dL/d image")
    close(ctrl.grad, G["cl_g_ctrl"], 1e-4, "dL/d control points")
    # the transposed-table fast path of the forward gives the same gradients
    P_hat_t = ops.transpose_p_hat(ph)
    img2, ctrl2 = img.detach().clone().requires_grad_(True), ctrl.detach().clone().requires_grad_(True)
    out2 = ops.warp_autograd(img2, ctrl2, inv, ph, cases.CL_HW, P_hat_t=P_hat_t, table_flags=ops.TABLE_MIRROR4)
    (out2 * dev(gi["g_out_cl"], cuda)).sum().backward()
    assert torch.equal(out2, out)
    close(ctrl2.grad, G["cl_g_ctrl"], 1e-4, "dL/d control points (transposed table)")


@pytest.mark.parametrize("transposed_score", [False, True])
def test_tpspp_backward_against_reference(cuda, transposed_score):
    G = cases.load("warp_backward")
    inp, gi = cases.g3_inputs(), cases.g14_inputs()
    c = O.tpspp_constants(cases.PP_HW, cases.PP_POINT)
    inv, ph, pxy = dev(c["hat_C"], cuda), dev(c["P_hat"], cuda), dev(c["P_xy"], cuda)
    fg = dev(inp["feat_grid"], cuda).requires_grad_(True)
    x = dev(inp["x"], cuda).requires_grad_(True)
    ctrl = dev(inp["ctrl"], cuda).requires_grad_(True)
    if transposed_score:      # the layout the module hands over: a transposed view of an (N, F, n) buffer
        base = dev(np.ascontiguousarray(inp["score"].transpose(0, 2, 1)), cuda).requires_grad_(True)
        score = base.transpose(1, 2)
    else:
        base = score = dev(inp["score"], cuda).requires_grad_(True)
    out0, out1 = ops.warp_autograd(fg, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, score=score, in1=x)
    ((out0 * dev(gi["g_out0"], cuda)).sum() + (out1 * dev(gi["g_out1"], cuda)).sum()).backward()
    g_score = base.grad.transpose(1, 2) if transposed_score else base.grad
    close(fg.grad[:, ::8], G["pp_g_feat_grid_sub"], 2e-5, "dL/d feat_grid")
    close(x.grad[:, ::8], G["pp_g_x_sub"], 2e-5, "dL/d x")
    close(ctrl.grad, G["pp_g_ctrl"], 1e-4, "dL/d control points")
    close(g_score, G["pp_g_score"], 1e-4, "dL/d score")


def test_backward_edge_cases_against_oracle(cuda):
    """Clamped coordinates (zero coordinate gradient), a single input, no score, ragged batch."""
    from tps_pp_amd import synth
    N, F, hw = 3, 20, (32, 100)
    c = O.classic_constants(F, hw)
    ctrl = O.classic_initial_ctrl(F)[None] + np.array([0.0, 0.6, 3.0], np.float32)[:, None, None] * \
        synth.dyadic((N, F, 2), "bwd.ctrl", 1)
    img = synth.smooth_image((N, 2, 16, 40), "bwd.img", 1)           # input size != output size
    g_out = synth.dyadic((N, 2) + hw, "bwd.gout", 1)
    # (grid from the FMA chain: at perturbation 3.0 the lattice is ill-conditioned and torch.bmm's own
    # result moves by 1e-4 with the BLAS kernel it picks for the batch size -- oracle/tps_oracle.py)
    want = O.warp_backward(g_out, img, ctrl, c["inv_delta_C"], c["P_hat"], hw, chain_grid=True)
    it, ct = dev(img, cuda).requires_grad_(True), dev(ctrl.astype(np.float32), cuda).requires_grad_(True)
    out = ops.warp_autograd(it, ct, dev(c["inv_delta_C"], cuda), dev(c["P_hat"], cuda), hw)
    (out * dev(g_out, cuda)).sum().backward()
    close(it.grad, want["g_in0"], 2e-5, "dL/d image (clamped grid)")
    close(ct.grad, want["g_ctrl"], 2e-4, "dL/d control points (clamped grid)")
    # only the control points need a gradient: the image-gradient buffer is not even allocated
    ct2 = ct.detach().clone().requires_grad_(True)
    (ops.warp_autograd(it.detach(), ct2, dev(c["inv_delta_C"], cuda), dev(c["P_hat"], cuda), hw) *
     dev(g_out, cuda)).sum().backward()
    close(ct2.grad, want["g_ctrl"], 2e-4, "dL/d control points (image detached)")


def test_modules_train_through_the_hip_warp(cuda):
    """TPS_PP / TPSPreprocessor under autograd: HIP warp forward + backward inside a PyTorch graph; the
    forward values equal the inference path's, and every parameter that feeds the warp gets a gradient."""
    m = TPS_PP().to(cuda).train()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    inp = cases.g4_inputs("ResNet45v2")
    x, outs = dev(inp["x"], cuda), [dev(o, cuda) for o in inp["outs"]]
    res = m(x, outs)
    assert res["output"].requires_grad and res["mp_img"].requires_grad
    (res["output"].square().mean() + res["mp_img"].square().mean()).backward()
    missing = [k for k, p in m.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    assert m.TPE.localization_fc2.bias.grad.abs().max() > 0
    with torch.no_grad():
        ref = m.eval()(x, outs)
    assert (ref["output"] - res["output"]).abs().max() < 1e-4

    p = TPSPreprocessor(20, (32, 100), (32, 100), 3).to(cuda).train()
    img = dev(cases.g1_inputs()["img"], cuda).requires_grad_(True)
    out = p(img)
    out.square().mean().backward()
    assert img.grad is not None and torch.isfinite(img.grad).all()
    assert p.LocalizationNetwork.localization_fc2.bias.grad.abs().max() > 0


def test_backward_full_size_properties(cuda):
    """TPS_PP geometry at batch 512 (the oracle cannot run this in seconds): the HIP backward against the
    HIP forward itself.  L = <out0, g0> + <out1, g1> is LINEAR in the sampled inputs, so
    <dL/d in, d> = L(in + d) - L(in) exactly (to rounding); the parameter gradients of a few rows are
    compared with float64 autograd of the reference's composition."""
    from tps_pp_amd import constants
    n = 512
    g = torch.Generator(device=cuda).manual_seed(5)
    c = O.tpspp_constants(cases.PP_HW, cases.PP_POINT)
    inv, ph, pxy = dev(c["hat_C"], cuda), dev(c["P_hat"], cuda), dev(c["P_xy"], cuda)
    # band-limited images (low-resolution noise, bilinearly upsampled): the loss is then smooth enough in
    # the control points for a finite difference to mean something
    up = lambda t, hw: torch.nn.functional.interpolate(t, size=hw, mode="bilinear", align_corners=True)  # noqa: E731
    fg = up(torch.rand((n, 64, 4, 16), generator=g, device=cuda), (32, 128)).contiguous()
    x = up(torch.rand((n, 64, 2, 8), generator=g, device=cuda), (16, 64)).contiguous()
    ctrl = dev(constants.tpspp_initial_ctrl((2, 16)), cuda)[None].repeat(n, 1, 1) + \
        0.02 * (torch.rand((n, 32, 2), generator=g, device=cuda) - 0.5)
    score = 0.5 * (torch.rand((n, 1024, 32), generator=g, device=cuda) - 0.5)
    g0 = torch.rand((n, 64, 16, 64), generator=g, device=cuda) - 0.5
    g1 = torch.rand((n, 64, 16, 64), generator=g, device=cuda) - 0.5

    def loss(fg_, x_, ctrl_, score_):
        o0, o1, _, _ = ops.warp(fg_, ctrl_, inv, ph, cases.PP_HW, P_xy=pxy, score=score_, in1=x_)
        return float((o0.double() * g0.double()).sum() + (o1.double() * g1.double()).sum())

    _, _, grid, _ = ops.warp(fg, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, score=score, in1=x, want_grid=True)
    g_fg, g_x, g_ctrl, g_score = ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, score=score,
                                                   in1=x, g_out1=g1)
    base = loss(fg, x, ctrl, score)
    # linear in the inputs
    d_fg = torch.rand(fg.shape, generator=g, device=cuda) - 0.5
    d_x = torch.rand(x.shape, generator=g, device=cuda) - 0.5
    want = loss(fg + d_fg, x + d_x, ctrl, score) - base
    got = float((g_fg.double() * d_fg.double()).sum() + (g_x.double() * d_x.double()).sum())
    assert abs(got - want) <= 2e-4 * max(abs(want), 1.0), (got, want)
    # control-point and score gradients of a few rows of the batch against float64 autograd of the
    # reference's composition (a finite difference of the fp32 forward is too noisy here: the sampled
    # function has a kink at every pixel boundary and wherever the grid leaves [-1, 1])
    import torch.nn.functional as Fn
    pick = [0, 1, 255, 511]
    k = len(pick)
    with torch.enable_grad():
        cd = ctrl[pick].cpu().double().requires_grad_(True)
        sd = score[pick].cpu().double().requires_grad_(True)
        rows = torch.cat([torch.ones(k, 1024, 1, dtype=torch.float64),
                          torch.from_numpy(c["P_xy"]).double()[None].repeat(k, 1, 1),
                          torch.from_numpy(c["P_hat"]).double()[None] * (sd * 0.5 + 1)], 2)
        T = torch.bmm(torch.from_numpy(c["hat_C"]).double()[None].repeat(k, 1, 1),
                      torch.cat((cd, torch.zeros(k, 3, 2, dtype=torch.float64)), 1))
        gr = torch.bmm(rows, T).reshape(k, 16, 64, 2)
        L = (Fn.grid_sample(fg[pick].cpu().double(), gr, padding_mode="border", align_corners=True) *
             g0[pick].cpu().double()).sum() + \
            (Fn.grid_sample(x[pick].cpu().double(), gr, padding_mode="border", align_corners=True) *
             g1[pick].cpu().double()).sum()
        L.backward()
    close(g_ctrl[pick], cd.grad.float().numpy(), 2e-4, "dL/d control points (rows of batch 512)")
    close(g_score[pick], sd.grad.float().numpy(), 2e-4, "dL/d score (rows of batch 512)")


def test_fixed_point_input_gradients_are_exact_and_reproducible(cuda):
    """TPS_PP geometry (<= 1024 output pixels): dL/d input is summed as round(w * g * 2^s) in 64-bit LDS integers.
    (i) two runs agree bit for bit (float atomics would not); (ii) against float64 autograd of the reference's
    sampler on the same fp32 grid the error is that of the fp32 tap weights (the coordinates are fp32 in the reference
    as well), not of the accumulation; (iii) the scale follows the data: gradients of magnitude 1e-30 and 1e+30 keep that
    relative accuracy; (iv) a non-finite incoming gradient poisons its own planes only."""
    import torch.nn.functional as Fn
    n = 6
    g = torch.Generator(device=cuda).manual_seed(9)
    c = O.tpspp_constants(cases.PP_HW, cases.PP_POINT)
    inv, ph, pxy = dev(c["hat_C"], cuda), dev(c["P_hat"], cuda), dev(c["P_xy"], cuda)
    from tps_pp_amd import constants
    fg = torch.rand((n, 5, 32, 128), generator=g, device=cuda)
    x = torch.rand((n, 3, 16, 64), generator=g, device=cuda)
    ctrl = dev(constants.tpspp_initial_ctrl((2, 16)), cuda)[None].repeat(n, 1, 1) + \
        0.3 * (torch.rand((n, 32, 2), generator=g, device=cuda) - 0.5)          # folds and clamps: many taps share pixels
    _, _, grid, _ = ops.warp(fg, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, in1=x, want_grid=True)
    for mag in (1.0, 1e-30, 1e30):
        g0 = (torch.rand((n, 5, 16, 64), generator=g, device=cuda) - 0.5) * mag
        g1 = (torch.rand((n, 3, 16, 64), generator=g, device=cuda) - 0.5) * mag
        a = ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, in1=x, g_out1=g1)
        b = ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, in1=x, g_out1=g1)
        assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
        with torch.enable_grad():
            fd, xd = fg.cpu().double().requires_grad_(True), x.cpu().double().requires_grad_(True)
            gr = grid.cpu().double().reshape(n, 16, 64, 2)
            L = (Fn.grid_sample(fd, gr, padding_mode="border", align_corners=True) * (g0.cpu().double() / mag)).sum() + \
                (Fn.grid_sample(xd, gr, padding_mode="border", align_corners=True) * (g1.cpu().double() / mag)).sum()
            L.backward()
        for got, want in ((a[0], fd.grad), (a[1], xd.grad)):
            got = got.cpu().double() / mag
            assert (got - want).abs().max() <= 2e-5 * want.abs().max(), (mag, float((got - want).abs().max()))
    g0 = torch.rand((n, 5, 16, 64), generator=g, device=cuda)
    g1 = torch.rand((n, 3, 16, 64), generator=g, device=cuda)
    g0[2, 3, 5, 7] = float("inf")
    r = ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, in1=x, g_out1=g1)
    assert torch.isnan(r[0][2, 3]).all() and torch.isfinite(r[0][2, :3]).all() and torch.isfinite(r[0][[0, 1, 3, 4, 5]]).all()
    assert torch.isfinite(r[1]).all()


def test_fixed_point_input_gradients_classic_geometry(cuda):
    """The classic 32x100 geometry takes the same fixed-point kernel with 1024-thread workgroups (3200 output pixels, four per
    thread): bitwise reproducible, and against float64 autograd of the sampler on the same grid within the fp32 tap weights'
    accuracy; odd channel count (the second pass has one plane), a batch that is not a multiple of anything."""
    import torch.nn.functional as Fn
    n, hw = 5, (32, 100)
    g = torch.Generator(device=cuda).manual_seed(11)
    c = O.classic_constants(20, hw)
    inv, ph = dev(c["inv_delta_C"], cuda), dev(c["P_hat"], cuda)
    img = torch.rand((n, 3, 32, 100), generator=g, device=cuda)
    ctrl = dev(c["C"].astype("float32"), cuda)[None].repeat(n, 1, 1) + 0.2 * (torch.rand((n, 20, 2), generator=g, device=cuda) - 0.5)
    _, _, grid, _ = ops.warp(img, ctrl, inv, ph, hw, want_grid=True)
    g0 = torch.rand((n, 3, 32, 100), generator=g, device=cuda) - 0.5
    a = ops.warp_backward(g0, img, grid, ctrl, inv, ph, hw)
    b = ops.warp_backward(g0, img, grid, ctrl, inv, ph, hw)
    assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32))
    with torch.enable_grad():
        fd = img.cpu().double().requires_grad_(True)
        (Fn.grid_sample(fd, grid.cpu().double().reshape(n, 32, 100, 2), padding_mode="border", align_corners=True) *
         g0.cpu().double()).sum().backward()
    err = (a[0].cpu().double() - fd.grad).abs().max()
    assert err <= 2e-5 * fd.grad.abs().max(), float(err)
