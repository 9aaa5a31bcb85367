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
    # dL/dC' = inv_delta_C^T P_hat^T dL/d grid cancels heavily (|inv_delta_C| up to ~220) and sums 3200 pixels: the
    # reference's own fp32 bmm backward is only good to ~1e-4 of the largest entry.  So: (i) the golden within 2e-4 (its
    # rounding + ours), and (ii) against the SAME chain in float64 on the forward's fp32 grid -- what both approximate --
    # ten times tighter (round 4: the reductions behind the per-thread partials run in fp64).
    close(ctrl.grad, G["cl_g_ctrl"], 2e-4, "dL/d control points")
    import torch.nn.functional as Fn
    with torch.no_grad():
        _, _, grid, _ = ops.warp(img.detach(), ctrl.detach(), inv, ph, cases.CL_HW, want_grid=True)
    n = ctrl.shape[0]
    with torch.enable_grad():
        gd = grid.cpu().double().reshape(n, cases.CL_HW[0], cases.CL_HW[1], 2).requires_grad_(True)
        (Fn.grid_sample(img.detach().cpu().double(), gd, padding_mode="border", align_corners=True) *
         torch.from_numpy(gi["g_out_cl"]).double()).sum().backward()
    gT = torch.matmul(torch.from_numpy(c["P_hat"]).double().t()[None], gd.grad.reshape(n, -1, 2))        # (n, K, 2)
    truth = torch.matmul(torch.from_numpy(c["inv_delta_C"]).double().t()[None], gT)[:, :cases.CL_F]
    close(ctrl.grad, truth.float().numpy(), 2e-5, "dL/d control points against float64 on the fp32 grid")
    assert np.abs(G["cl_g_ctrl"] - truth.numpy()).max() <= 2e-4 * np.abs(truth.numpy()).max()
    # the transposed-table fast path of the forward gives the same gradients
    P_hat_t = ops.transpose_p_hat(ph)
    img2, ctrl2 = img.detach().clone().requires_grad_(True), ctrl.detach().clone().requires_grad_(True)
    out2 = ops.warp_autograd(img2, ctrl2, inv, ph, cases.CL_HW, P_hat_t=P_hat_t, table_flags=ops.TABLE_MIRROR4)
    (out2 * dev(gi["g_out_cl"], cuda)).sum().backward()
    assert torch.equal(out2, out)
    close(ctrl2.grad, truth.float().numpy(), 2e-5, "dL/d control points (transposed table)")


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


def test_fixed_point_input_gradients_are_exact_and_reproducible(cuda, request):
    """(The fixed-point accumulator of round 3, selected per call with `ops.warp_backward(..., fixed_point=True)` = TPSPP_BWD_FIXED_POINT.)  TPS_PP geometry (<= 1024 output pixels): dL/d input is summed as round(w * g * 2^s) in 64-bit LDS integers.
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
        a = ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, in1=x, g_out1=g1, fixed_point=True)
        b = ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, in1=x, g_out1=g1, fixed_point=True)
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
    r = ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, in1=x, g_out1=g1, fixed_point=True)
    assert torch.isnan(r[0][2, 3]).all() and torch.isfinite(r[0][2, :3]).all() and torch.isfinite(r[0][[0, 1, 3, 4, 5]]).all()
    assert torch.isfinite(r[1]).all()


def test_fixed_point_input_gradients_classic_geometry(cuda, request):
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
    a = ops.warp_backward(g0, img, grid, ctrl, inv, ph, hw, fixed_point=True)
    b = ops.warp_backward(g0, img, grid, ctrl, inv, ph, hw, fixed_point=True)
    assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32))
    with torch.enable_grad():
        fd = img.cpu().double().requires_grad_(True)
        (Fn.grid_sample(fd, grid.cpu().double().reshape(n, 32, 100, 2), padding_mode="border", align_corners=True) *
         g0.cpu().double()).sum().backward()
    err = (a[0].cpu().double() - fd.grad).abs().max()
    assert err <= 2e-5 * fd.grad.abs().max(), float(err)


def test_default_accumulator_run_to_run_difference_is_bounded(cuda):
    """The default dL/d input accumulator is an fp64 LDS atomic sum: its value depends on the order in which the atomics
    arrive (2^-53 relative per add), so after the rounding to fp32 two runs may differ -- by at most ONE fp32 ulp and only at
    rounding ties (include/tpspp.h says so).  Six runs on a fold-heavy grid (many taps share a pixel), the per-call
    fixed-point mode beside them as the reproducible alternative: (i) every run within one ulp of the first, element by
    element; (ii) the fixed-point call bit-identical twice, and within the documented accuracy of the default;
    (iii) dL/d control points and dL/d score identical in every run (fixed summation order in either mode)."""
    n = 8
    g = torch.Generator(device=cuda).manual_seed(21)
    c = O.tpspp_constants(cases.PP_HW, cases.PP_POINT)
    inv, ph, pxy = dev(c["hat_C"], cuda), dev(c["P_hat"], cuda), dev(c["P_xy"], cuda)
    from tps_pp_amd import constants
    fg = torch.rand((n, 6, 32, 128), generator=g, device=cuda)
    x = torch.rand((n, 4, 16, 64), generator=g, device=cuda)
    ctrl = dev(constants.tpspp_initial_ctrl((2, 16)), cuda)[None].repeat(n, 1, 1) + \
        0.3 * (torch.rand((n, 32, 2), generator=g, device=cuda) - 0.5)
    sc = torch.tanh(torch.randn((n, 1024, 32), generator=g, device=cuda))
    _, _, grid, _ = ops.warp(fg, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, score=sc, in1=x, want_grid=True)
    g0 = torch.randn((n, 6, 16, 64), generator=g, device=cuda)
    g1 = torch.randn((n, 4, 16, 64), generator=g, device=cuda)

    def run(**kw):
        return ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, score=sc, in1=x, g_out1=g1, **kw)

    def ulps(a, b):
        """Distance in units of the last place (both finite, same sign or zero)."""
        ia, ib = a.contiguous().view(torch.int32).long(), b.contiguous().view(torch.int32).long()
        ia = torch.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
        ib = torch.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
        return (ia - ib).abs()

    first = run()
    worst = 0
    for _ in range(5):
        r = run()
        for k in (0, 1):
            worst = max(worst, int(ulps(first[k], r[k]).max()))
        assert torch.equal(first[2], r[2]) and torch.equal(first[3], r[3])
    assert worst <= 1, f"default accumulator: runs differ by {worst} ulp"
    fa, fb = run(fixed_point=True), run(fixed_point=True)
    for k in (0, 1):
        assert torch.equal(fa[k].view(torch.int32), fb[k].view(torch.int32))
        tol = 4e-7 * float(first[k].abs().max())                    # fixed point: exact within 2^-50 of the pass's largest |g|
        assert float((fa[k] - first[k]).abs().max()) <= tol
    assert torch.equal(fa[2], first[2]) and torch.equal(fa[3], first[3])


def _scatter_reference(grid, g_out, H, W):
    """dL/d input of the bilinear sampler as the EXACT sum (float64) of the kernel's own fp32 terms w * g: tap weights
    and offsets in fp32 with the kernel's operation order (tpspp_warp_bwd.hip, = ATen's), every product rounded to
    fp32, the accumulation in float64.  Returns (sum, sum of |terms|) per input element."""
    f = np.float32
    n, C, Ho, Wo = g_out.shape
    gx, gy = grid[..., 0].reshape(n, -1).astype(f), grid[..., 1].reshape(n, -1).astype(f)
    ix = ((gx + f(1)) * f(0.5)) * f(W - 1)
    iy = ((gy + f(1)) * f(0.5)) * f(H - 1)
    ix = np.where(ix <= 0, f(0), np.where(ix >= f(W - 1), f(W - 1), ix)).astype(f)
    iy = np.where(iy <= 0, f(0), np.where(iy >= f(H - 1), f(H - 1), iy)).astype(f)
    fx, fy = np.floor(ix), np.floor(iy)
    x0, y0 = fx.astype(np.int64), fy.astype(np.int64)
    w, nn = (ix - fx).astype(f), (iy - fy).astype(f)
    e, s_ = (f(1) - w).astype(f), (f(1) - nn).astype(f)
    wts = [(s_ * e).astype(f), (s_ * w).astype(f), (nn * e).astype(f), (nn * w).astype(f)]
    inx, iny = (x0 + 1) < W, (y0 + 1) < H
    oks = [np.ones_like(inx), inx, iny, inx & iny]
    offs = [y0 * W + x0, y0 * W + x0 + 1, (y0 + 1) * W + x0, (y0 + 1) * W + x0 + 1]
    tot = np.zeros((n, C, H * W), np.float64)
    mass = np.zeros((n, C, H * W), np.float64)
    go = g_out.reshape(n, C, -1).astype(f)
    for b in range(n):
        for c in range(C):
            for wt, ok, of in zip(wts, oks, offs):
                term = (wt[b] * go[b, c]).astype(f)                # the kernel's fp32 product
                sel = ok[b] & np.isfinite(term)
                np.add.at(tot[b, c], of[b][sel], term[sel].astype(np.float64))
                np.add.at(mass[b, c], of[b][sel], np.abs(term[sel]).astype(np.float64))
    return tot.reshape(n, C, H, W), mass.reshape(n, C, H, W)


@pytest.mark.parametrize("geometry", ["tpspp", "classic"])
def test_input_gradients_mixed_magnitudes(cuda, geometry):
    """Gradients log-uniform over 1e-8 .. 1e2 in every plane, plus one 1e3 outlier in one of them: every element of
    dL/d input must be the fp32 rounding of the exact sum of its own fp32 terms w * g (better than any order of fp32
    atomics), however small the element is next to the plane's largest gradient.
    (The fixed-point accumulator of round 3 fails this: its scale follows the largest |g| of a pass.)  Also: a non-finite
    incoming gradient poisons the taps it touches and nothing else, as ATen's kernel does."""
    g = torch.Generator(device=cuda).manual_seed(21)
    if geometry == "tpspp":
        n, hw = 4, cases.PP_HW
        c = O.tpspp_constants(cases.PP_HW, cases.PP_POINT)
        inv, ph, pxy = dev(c["hat_C"], cuda), dev(c["P_hat"], cuda), dev(c["P_xy"], cuda)
        from tps_pp_amd import constants
        inp = torch.rand((n, 5, 32, 128), generator=g, device=cuda)
        ctrl = dev(constants.tpspp_initial_ctrl((2, 16)), cuda)[None].repeat(n, 1, 1) + \
            0.2 * (torch.rand((n, 32, 2), generator=g, device=cuda) - 0.5)
        kw = dict(P_xy=pxy)
    else:
        n, hw = 3, (32, 100)
        c = O.classic_constants(20, hw)
        inv, ph = dev(c["inv_delta_C"], cuda), dev(c["P_hat"], cuda)
        inp = torch.rand((n, 3, 32, 100), generator=g, device=cuda)
        ctrl = dev(c["C"].astype("float32"), cuda)[None].repeat(n, 1, 1) + 0.2 * (torch.rand((n, 20, 2), generator=g, device=cuda) - 0.5)
        kw = {}
    _, _, grid, _ = ops.warp(inp, ctrl, inv, ph, hw, want_grid=True, **kw)
    C = inp.shape[1]
    expo = torch.rand((n, C) + tuple(hw), generator=g, device=cuda) * 10.0 - 8.0
    sign = torch.where(torch.rand(expo.shape, generator=g, device=cuda) < 0.5, -1.0, 1.0)
    g0 = sign * torch.pow(10.0, expo)
    g0[1, 2, 3, 5] = 1.0e3                                      # the outlier
    got = ops.warp_backward(g0, inp, grid, ctrl, inv, ph, hw, **kw)[0].cpu().double().numpy()
    H, W = inp.shape[2:]
    want, mass = _scatter_reference(grid.cpu().numpy().reshape((n,) + tuple(hw) + (2,)), g0.cpu().numpy(), H, W)
    err = np.abs(got - want)
    # the fp32 rounding of the exact sum, plus the fp64 accumulation's own rounding (a few 2^-53 of the terms' mass)
    bound = 2.0 ** -24 * np.abs(want) + 1e-14 * mass + 1e-45
    worst = float((err / bound).max())
    assert worst <= 1.0, f"{geometry}: an element is off by {worst:.2f}x the rounding of its own exact sum"
    assert float((mass > 0).mean()) > 0.25                     # the test did exercise a good part of the planes
    # the smallest elements really are far below the plane's maximum (what a per-pass scale cannot resolve)
    pl = np.abs(want[1, 2])
    assert float(pl[pl > 0].min()) < 1e-9 * float(pl.max())
    # non-finite gradient: its four taps, nothing else
    g1 = g0.clone()
    g1[0, 1, 4, 9] = float("inf")
    r = ops.warp_backward(g1, inp, grid, ctrl, inv, ph, hw, **kw)[0]
    bad = ~torch.isfinite(r)
    assert 1 <= int(bad.sum()) <= 4 and bool(bad[0, 1].any()) and int(bad.sum()) == int(bad[0, 1].sum())


@pytest.mark.parametrize("C,hw,n", [(3, (32, 100), 37), (1, (32, 100), 5), (2, (32, 64), 9), (3, (24, 132), 6), (3, (48, 64), 4)])
def test_classic_backward_single_launch_against_the_two_kernel_route(cuda, C, hw, n):
    """Round 5: the classic rectifier's backward (one input of <= 3 channels, no score, 1024 < pixels <= 4096, transposed
    table) is ONE launch (warp_bwd_classic_kernel: image staged in LDS, fp64 LDS accumulators, dL/dT finished in the
    workgroup); TPSPP_BWD_TWO_KERNELS=1 selects the sampling + parameter kernels of rounds 3-5.  Same sampling arithmetic:
    dL/d input within one ulp (fp64 sums of the same fp32 terms -- neighbouring lanes add their shared tap in fp64 before
    the atomic, so only the order of the fp64 additions differs); dL/dC' within 5e-5 of the largest entry of the two-kernel
    route (which keeps fp32 partial sums of ~12 terms per lane: measured 1.8e-5 apart) and within 3e-5 of float64 on the
    same grid (measured 1.5e-5: the fp32 coordinate gradients of the sampler, amplified by |inv_delta_C| ~ 220).  From gentle to strong deformations (folds, coordinates clamped
    at the borders), a NaN in the incoming gradient (stays in its four taps; that image's dL/dC' is NaN in both routes)."""
    import torch.nn.functional as Fn
    from tps_pp_amd import constants
    g = torch.Generator(device=cuda).manual_seed(17 + C + hw[1])
    p = TPSPreprocessor(20, hw, hw, C).eval().to(cuda)
    gg = p.GridGenerator
    P_hat_t, flags = gg.prepared_table()
    img = torch.rand((n, C) + hw, generator=g, device=cuda)
    ctrl = dev(constants.classic_initial_ctrl(20), cuda)[None].repeat(n, 1, 1).contiguous()
    amp = torch.linspace(0.02, 0.9, n, device=cuda)[:, None, None]          # from gentle to folded / clamped
    ctrl = ctrl + amp * (torch.rand(ctrl.shape, generator=g, device=cuda) - 0.5)
    go = torch.randn((n, C) + hw, generator=g, device=cuda)
    go[0, 0, 3, 7] = float("nan")
    _, _, grid, _ = ops.warp(img, ctrl, gg.inv_delta_C, gg.P_hat, hw, want_grid=True, P_hat_t=P_hat_t, table_flags=flags)

    def run(two_kernels=False):
        return ops.warp_backward(go, img, grid, ctrl, gg.inv_delta_C, gg.P_hat, hw, P_hat_t=P_hat_t, two_kernels=two_kernels)

    want = run(two_kernels=True)                 # (the per-call TPSPP_BWD_TWO_KERNELS bit; the environment variable is read once)
    for rep in range(3):
        got = run()
        gi, wi = got[0], want[0]
        assert torch.equal(torch.isnan(gi), torch.isnan(wi)) and int(torch.isnan(gi).sum()) in range(1, 5)
        ok = ~torch.isnan(wi)
        ia, ib = gi[ok].view(torch.int32).long(), wi[ok].view(torch.int32).long()
        ia = torch.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
        ib = torch.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
        assert int((ia - ib).abs().max()) <= 1, ("dL/d input", C, hw, rep, int((ia - ib).abs().max()))
        # image 0 carries the NaN: its dL/dC' is NaN in both routes; the others are finite
        assert torch.isnan(got[2][0]).all() == torch.isnan(want[2][0]).all()
        close(got[2][1:], want[2][1:].cpu().numpy(), 5e-5, "dL/d control points (single launch against two kernels)")
    # dL/d input not wanted (the kernel then skips the accumulators): the same dL/dC', bit for bit
    g_no = ops.warp_backward(go, img, grid, ctrl, gg.inv_delta_C, gg.P_hat, hw, P_hat_t=P_hat_t, need_in0=False)
    assert g_no[0] is None and torch.equal(g_no[2][1:], got[2][1:])
    # float64 on the same fp32 grid, images without the NaN
    with torch.enable_grad():
        gd = grid[1:].cpu().double().reshape(n - 1, hw[0], hw[1], 2).requires_grad_(True)
        (Fn.grid_sample(img[1:].cpu().double(), gd, padding_mode="border", align_corners=True) * go[1:].cpu().double()).sum().backward()
    gT = torch.matmul(gg.P_hat.cpu().double().t()[None], gd.grad.reshape(n - 1, -1, 2))
    truth = torch.matmul(gg.inv_delta_C.cpu().double().t()[None], gT)[:, :20]
    close(got[2][1:], truth.float().numpy(), 3e-5, "dL/d control points against float64 on the fp32 grid")


def test_backward_workspace_size_is_checked(cuda):
    """The C entry point refuses a workspace smaller than tpspp_warp_bwd_workspace_floats() (round 3 grew it silently;
    since ABI version 2 the size travels with the pointer)."""
    import ctypes
    from tps_pp_amd import _lib
    L = _lib.lib()
    n, hw = 2, (32, 100)
    c = O.classic_constants(20, hw)
    inv, ph = dev(c["inv_delta_C"], cuda), dev(c["P_hat"], cuda)
    img = torch.rand((n, 3, 32, 100), device=cuda)
    ctrl = dev(c["C"].astype("float32"), cuda)[None].repeat(n, 1, 1).contiguous()
    _, _, grid, _ = ops.warp(img, ctrl, inv, ph, hw, want_grid=True)
    T = ops.solve_T(inv, ctrl)
    g0 = torch.rand_like(img)
    g_in, g_ctrl = torch.empty_like(img), torch.empty((n, 20, 2), device=cuda)
    need = int(L.tpspp_warp_bwd_workspace_floats(n, 32, 100))
    ws = torch.empty((need,), device=cuda)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())   # noqa: E731

    def call(floats):
        return L.tpspp_warp_bwd(P(g0), P(img), 3, 32, 100, P(None), P(None), 0, 0, 0, P(grid), P(T), P(inv), P(ph), 23,
                                P(None), P(None), P(None), 0, n, 20, 32, 100, P(g_in), P(None), P(g_ctrl), P(None),
                                P(ws), floats, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert call(n * 3200 * 2) == -22            # the documented size of rounds 1-2: refused, nothing launched
    assert call(need) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(g_ctrl).all() and torch.isfinite(g_in).all()
