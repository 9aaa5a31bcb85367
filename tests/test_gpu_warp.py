"""-m gpu parity tests of the warp hot path, through the C ABI (tps_pp_amd.ops -> libtpspp_hip.so).

Bars (BASELINE.json north_star): sampling grid and corner indices BIT-EXACT, warped tensors within
1e-4 of the reference -- in fact every comparison below is bit-for-bit, against
  (a) the committed golden outputs of the reference itself (tests/golden/*.npz), and
  (b) the CPU oracle on fresh seeded inputs, including ragged / edge shapes.
"""
import numpy as np
import pytest
import torch

import cases
from tps_pp_amd import _lib, ops, synth

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north-star tolerance for warped values (we assert 0 where noted)


def dev(a, cuda):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_biteq(a, b, what):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype == np.float32:
        ne = bits(a) != bits(b)
    else:
        ne = a != b
    assert not ne.any(), f"{what}: {int(ne.sum())} of {ne.size} elements differ, " \
                         f"max abs {np.abs(a.astype(np.float64) - b.astype(np.float64)).max():.3e}"


# ---------------------------------------------------------------- golden: classic (G2) -----------
def test_classic_warp_matches_reference_golden(cuda):
    K = cases.load("constants")
    G = cases.load("classic_warp")
    inp = cases.g2_inputs()
    inv, P_hat = dev(K["classic_inv_delta_C"], cuda), dev(K["classic_P_hat"], cuda)
    P_hat_t = ops.transpose_p_hat(P_hat)
    assert torch.equal(P_hat_t, P_hat.t().contiguous())
    assert ops.table_mirror_symmetry(K["classic_P_hat"], cases.CL_HW, cases.CL_F) == 1
    P_prep, packed = ops.prepare_mirror_table(P_hat, cases.CL_HW)
    assert packed == ops.TABLE_PACKED | ops.TABLE_SPAN and torch.equal(P_prep, P_hat_t)
    try:
        # 1 = gather, 3 = LDS-staged kernel, 2 = LDS-staged kernel on the mirror-symmetric table,
        # 5 = image-pair kernel, 6 = in-place kernel (both: prepared table), 0 = whatever the library picks for a prepared table
        for kernel, bands in ((1, 0), (3, 1), (3, 2), (3, 3), (2, 1), (2, 2), (2, 3), (5, 0), (6, 0), (0, 0)):
            ops.set_warp_tuning(0, 0, kernel, bands)
            tab, flags = (P_prep, ops.TABLE_MIRROR4 | packed) if kernel in (5, 6, 0) else (P_hat_t, ops.TABLE_MIRROR4)
            for key_in, key_out in (("img", "out"), ("img_smooth", "out_smooth")):
                out, _, grid, idx = ops.warp(dev(inp[key_in], cuda), dev(inp["ctrl"], cuda), inv,
                                             P_hat, cases.CL_HW, want_grid=True, want_idx=True,
                                             P_hat_t=tab, table_flags=flags)
                assert_biteq(grid, G["grid"], f"kernel {kernel}: grid vs reference bmm")
                assert np.abs(out.cpu().numpy() - G[key_out]).max() <= TOL
                assert_biteq(out, G[key_out], f"kernel {kernel}: warped {key_in} vs reference")
                # the same call without the optional outputs takes another instantiation of the kernels
                out2, _, _, _ = ops.warp(dev(inp[key_in], cuda), dev(inp["ctrl"], cuda), inv, P_hat, cases.CL_HW,
                                         P_hat_t=tab, table_flags=flags)
                assert_biteq(out2, G[key_out], f"kernel {kernel}: warped {key_in}, no grid / idx outputs")
    finally:
        ops.set_warp_tuning(0, 0, 0)


def test_packed_table_contract(cuda):
    """TABLE_PACKED is a promise about the memory behind P_hat_t: the binding refuses anything but the view that
    prepare_mirror_table returned; a geometry without a prepared form falls back to the plain transposition."""
    K = cases.load("constants")
    P_hat = dev(K["classic_P_hat"], cuda)
    inv = dev(K["classic_inv_delta_C"], cuda)
    inp = cases.g2_inputs()
    with pytest.raises(ValueError):
        ops.warp(dev(inp["img"], cuda), dev(inp["ctrl"], cuda), inv, P_hat, cases.CL_HW,
                 P_hat_t=ops.transpose_p_hat(P_hat), table_flags=ops.TABLE_MIRROR4 | ops.TABLE_PACKED)
    with pytest.raises(ValueError):
        ops.warp(dev(inp["img"], cuda), dev(inp["ctrl"], cuda), inv, P_hat, cases.CL_HW, table_flags=ops.TABLE_PACKED)
    odd = torch.zeros((31 * 99, 23), device=cuda)
    t, flags = ops.prepare_mirror_table(odd, (31, 99))
    assert flags == 0 and tuple(t.shape) == (23, 31 * 99)
    # the packed copy itself: thread t of the image-pair kernel = (r, c) of a 4-column x 8-row block per half-wavefront
    P_prep, packed = ops.prepare_mirror_table(P_hat, cases.CL_HW)
    tail = P_prep._base[23 * 3200:].cpu().numpy().reshape(13, 6, 64, 4)
    ref = np.zeros((13, 6, 64, 4), np.float32)
    ph = K["classic_P_hat"]
    for t_ in range(832):
        hw, l5 = t_ >> 5, t_ & 31
        rg, cg = divmod(hw, 13)
        r, c = rg * 8 + (l5 >> 2), cg * 4 + (l5 & 3)
        row = np.zeros(24, np.float32)
        row[:23] = ph[r * 100 + c]
        ref[t_ // 64, :, t_ % 64, :] = row.reshape(6, 4)
    assert (bits(tail) == bits(ref)).all()
    # 32x128 (a row pitch that is a multiple of 32 banks): a half-wavefront owns 32 columns of ONE row, a thread two rows
    # -> 2 column groups x 8 row-group pairs = 16 half-wavefronts, [8 wavefronts][2][6][64][4]
    from oracle import tps_oracle as O
    ph128 = O.classic_constants(20, (32, 128))["P_hat"]
    p128, f128 = ops.prepare_mirror_table(dev(ph128, cuda), (32, 128))
    assert f128 == ops.TABLE_PACKED | ops.TABLE_SPAN
    tail = p128._base[23 * 4096:].cpu().numpy().reshape(8, 2, 6, 64, 4)
    ref = np.zeros((8, 2, 6, 64, 4), np.float32)
    for t_ in range(512):
        hw, l5 = t_ >> 5, t_ & 31
        rgb, cg = divmod(hw, 2)
        for j in range(2):
            r, c = rgb * 2 + j, cg * 32 + l5
            row = np.zeros(24, np.float32)
            row[:23] = ph128[r * 128 + c]
            ref[t_ // 64, j, :, t_ % 64, :] = row.reshape(6, 4)
    assert (bits(tail) == bits(ref)).all()
    # shape / device / dtype of caller-supplied outputs are checked before any pointer reaches a kernel
    img, ctrl = dev(inp["img"], cuda), dev(inp["ctrl"], cuda)
    for bad in (torch.empty((img.shape[0], 3, 32, 99), device=cuda), torch.empty(img.shape, device=cuda, dtype=torch.bfloat16),
                torch.empty(img.shape), torch.empty((img.shape[0] - 1, 3, 32, 100), device=cuda)):
        with pytest.raises((ValueError, TypeError, _lib.TpsppError)):
            ops.warp(img, ctrl, inv, P_hat, cases.CL_HW, out0=bad)
        with pytest.raises((ValueError, TypeError, _lib.TpsppError)):
            ops.WarpPlan(img, ctrl, inv, P_hat, cases.CL_HW, bad)
    with pytest.raises(ValueError):
        ops.WarpPlan(img, ctrl, inv, P_hat[:, :20].contiguous(), cases.CL_HW, torch.empty_like(img),
                     P_xy=torch.zeros((5, 2), device=cuda))       # P_xy must be (n, 2)


def test_unfused_pieces_match_reference_golden(cuda):
    K = cases.load("constants")
    G = cases.load("classic_warp")
    inp = cases.g2_inputs()
    T = ops.solve_T(dev(K["classic_inv_delta_C"], cuda), dev(inp["ctrl"], cuda))
    grid = ops.build_grid(dev(K["classic_P_hat"], cuda), T)
    assert_biteq(grid, G["grid"], "build_grid")
    out, idx = ops.grid_sample(dev(inp["img"], cuda),
                               grid.reshape(cases.CL_N, *cases.CL_HW, 2), return_idx=True)
    assert_biteq(out, G["out"], "grid_sample")


# ---------------------------------------------------------------- golden: TPS_PP warp (G3) -------
def test_tpspp_warp_matches_reference_golden(cuda):
    K = cases.load("constants")
    G = cases.load("tpspp_warp")
    inp = cases.g3_inputs()
    P_xy = K["pp_P"].astype(np.float32)
    P_hat = dev(K["pp_P_hat"], cuda)
    try:
        for kernel, P_hat_t in ((1, None), (4, None), (4, ops.transpose_p_hat(P_hat))):
            ops.set_warp_tuning(0, 0, kernel, 0)      # 1 = gather kernel, 4 = plane-streaming kernel
            out0, out1, grid, _ = ops.warp(dev(inp["feat_grid"], cuda), dev(inp["ctrl"], cuda),
                                           dev(K["pp_hat_C"], cuda), P_hat, cases.PP_HW,
                                           P_xy=dev(P_xy, cuda), score=dev(inp["score"], cuda),
                                           in1=dev(inp["x"], cuda), want_grid=True, P_hat_t=P_hat_t)
            assert_biteq(grid, G["grid"], f"TPS_PP grid vs reference (kernel {kernel})")
            assert_biteq(out0, G["output"], f"TPS_PP output (kernel {kernel})")
            assert_biteq(out1, G["mp_img"], f"TPS_PP mp_img (kernel {kernel})")
    finally:
        ops.set_warp_tuning(0, 0, 0, 0)


def test_tpspp_warp_bf16_io(cuda, oracle):
    """TPSPP_IO_BF16: bf16 planes in and out, fp32 grid and interpolation.  On bf16-valued inputs the result
    must be the oracle's fp32 result rounded once to bfloat16 (nearest even), bit for bit; the grid is the
    fp32 one.  Odd batch, with and without the second input / score / transposed table."""
    K = cases.load("constants")
    G = cases.load("tpspp_warp")
    inp = cases.g3_inputs()
    P_xy = K["pp_P"].astype(np.float32)
    P_hat = dev(K["pp_P_hat"], cuda)
    rb = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16)      # noqa: E731
    fg, x = rb(inp["feat_grid"]), rb(inp["x"])
    ref = oracle.warp(fg.float().numpy(), inp["ctrl"], K["pp_hat_C"], K["pp_P_hat"], cases.PP_HW, P_xy=P_xy,
                      score=inp["score"], in1=x.float().numpy(), want_grid=True)
    for P_hat_t in (None, ops.transpose_p_hat(P_hat)):
        out0, out1, grid, _ = ops.warp(fg.to(cuda), dev(inp["ctrl"], cuda), dev(K["pp_hat_C"], cuda), P_hat,
                                       cases.PP_HW, P_xy=dev(P_xy, cuda), score=dev(inp["score"], cuda),
                                       in1=x.to(cuda), want_grid=True, P_hat_t=P_hat_t)
        assert out0.dtype == torch.bfloat16 and out1.dtype == torch.bfloat16
        assert_biteq(grid, G["grid"], "grid (fp32, as without the flag)")
        assert torch.equal(out0.cpu(), torch.from_numpy(ref["out0"]).to(torch.bfloat16))
        assert torch.equal(out1.cpu(), torch.from_numpy(ref["out1"]).to(torch.bfloat16))
    # single input, no score, one image
    ref1 = oracle.warp(x.float().numpy()[:1], inp["ctrl"][:1], K["pp_hat_C"], K["pp_P_hat"], cases.PP_HW, P_xy=P_xy)
    o1 = ops.warp(x[:1].to(cuda), dev(inp["ctrl"][:1], cuda), dev(K["pp_hat_C"], cuda), P_hat, cases.PP_HW,
                  P_xy=dev(P_xy, cuda))[0]
    assert torch.equal(o1.cpu(), torch.from_numpy(ref1["out0"]).to(torch.bfloat16))
    with pytest.raises(TypeError):
        ops.warp(fg.to(cuda), dev(inp["ctrl"], cuda), dev(K["pp_hat_C"], cuda), P_hat, cases.PP_HW,
                 P_xy=dev(P_xy, cuda), in1=dev(inp["x"], cuda))                   # mixed dtypes
    from tps_pp_amd import _lib
    with pytest.raises(_lib.TpsppError):                                          # classic 32x100: 3200 output pixels
        ops.warp(torch.zeros((2, 3, 32, 100), device=cuda, dtype=torch.bfloat16), torch.zeros((2, 20, 2), device=cuda),
                 torch.zeros((23, 23), device=cuda), torch.zeros((3200, 23), device=cuda), (32, 100))


# ---------------------------------------------------------------- oracle on fresh inputs ---------
@pytest.mark.parametrize("N,C,H,W,Ho,Wo,F,perturb", [
    (1, 1, 32, 100, 32, 100, 20, 0.05),     # the reference's own test shape (test_ocr_preprocessor.py:19-29)
    (5, 3, 32, 100, 32, 100, 20, 0.3),
    (3, 3, 32, 128, 32, 128, 20, 0.1),      # nrtr_tps++.py's commented-out preprocessor geometry
    (7, 2, 17, 33, 9, 21, 6, 0.5),          # ragged: generic-F kernel, odd sizes, tile tail
    (2, 5, 8, 8, 40, 70, 10, 2.0),          # upsampling warp, mostly clamped to the border
    (33, 3, 32, 100, 32, 100, 20, 0.05),    # > one image group
    (2, 3, 2, 2, 4, 4, 4, 1.0),             # tiny planes
    (2, 3, 1, 64, 8, 64, 8, 0.2),           # H == 1: y scale is 0
    (9, 1, 32, 100, 32, 100, 20, 0.2),      # odd batch, 1 channel: LDS kernel with a lone last image
    (4, 3, 48, 160, 48, 160, 20, 0.1),      # pair = 180 KB: too big for the pair kernel
    (6, 3, 16, 64, 16, 64, 20, 0.3),        # small planes: LDS kernel with 2 pixel slots
    (3, 3, 32, 100, 31, 99, 20, 0.2),       # odd output size: table is not mirror-symmetric
    (3, 3, 32, 160, 32, 160, 20, 0.1),      # the reference's recog-config test shape (test_recog_config.py:103-157): run-time-geometry in-place kernel
    (2, 1, 64, 256, 64, 256, 20, 0.1),      # ... 64 KB plane, two workgroups ("bands") per image
    (3, 4, 48, 160, 48, 160, 20, 0.3),      # ... four channels, 123 KB per image
    (5, 3, 64, 200, 64, 200, 20, 0.6),      # ... W % 32 = 8: pixel blocks of 8 x 4; clamped and out-of-image taps
])
def test_classic_warp_vs_oracle(cuda, oracle, N, C, H, W, Ho, Wo, F, perturb):
    Kc = oracle.classic_constants(F, (Ho, Wo))
    ctrl = oracle.classic_initial_ctrl(F)[None] + perturb * synth.dyadic((N, F, 2), "t.ctrl", N)
    img = synth.dyadic((N, C, H, W), "t.img", N + 1)
    ref = oracle.warp(img, ctrl, Kc["inv_delta_C"], Kc["P_hat"], (Ho, Wo), want_grid=True,
                      want_idx=True)
    P_hat = dev(Kc["P_hat"], cuda)
    sym = ops.table_mirror_symmetry(Kc["P_hat"], (Ho, Wo), F)
    assert sym == (1 if (Ho % 2 == 0 and Wo % 2 == 0 and F % 2 == 0) else 0)
    # generic path / coalesced + LDS paths / mirror-symmetric-table path / prepared table (image-pair kernel
    # where the geometry has one: odd batches end in a group with a single image)
    prep, packed = ops.prepare_mirror_table(P_hat, (Ho, Wo))
    for P_hat_t, flags in ((None, 0), (ops.transpose_p_hat(P_hat), 0),
                           (ops.transpose_p_hat(P_hat), ops.TABLE_MIRROR4 * sym),
                           (prep, ops.TABLE_MIRROR4 * sym | packed)):
        out, _, grid, idx = ops.warp(dev(img, cuda), dev(ctrl, cuda), dev(Kc["inv_delta_C"], cuda),
                                     P_hat, (Ho, Wo), want_grid=True, want_idx=True,
                                     P_hat_t=P_hat_t, table_flags=flags)
        assert_biteq(grid, ref["grid"], "grid")
        assert_biteq(idx, ref["idx"], "corner indices")
        assert_biteq(out, ref["out0"], "warped")


@pytest.mark.parametrize("N,point,hw,C0,C1,with_score", [
    (3, (2, 16), (16, 64), 64, 64, True),
    (2, (2, 16), (16, 64), 64, 64, False),
    (4, (2, 8), (8, 32), 6, 3, True),       # generic-F path with score + second input
    (1, (2, 3), (4, 10), 2, 1, True),
    (5, (2, 16), (16, 64), 7, 3, True),     # unequal channel counts: one input runs out of planes
    (2, (2, 16), (16, 64), 1, 5, True),
    (3, (2, 10), (10, 50), 4, 4, True),     # F = 20 in the TPS_PP layout, planes not KB-multiples
    (2, (2, 16), (8, 64), 3, 2, False),     # one pixel per thread
])
def test_tpspp_warp_vs_oracle(cuda, oracle, N, point, hw, C0, C1, with_score):
    Kp = oracle.tpspp_constants(hw, point)
    F = point[0] * point[1]
    n = hw[0] * hw[1]
    ctrl = oracle.tpspp_initial_ctrl(point)[None] + 0.1 * synth.dyadic((N, F, 2), "p.ctrl", N)
    score = synth.dyadic((N, n, F), "p.score", N) if with_score else None
    in0 = synth.dyadic((N, C0, 2 * hw[0], 2 * hw[1]), "p.in0", N)
    in1 = synth.dyadic((N, C1) + tuple(hw), "p.in1", N)
    ref = oracle.warp(in0, ctrl, Kp["hat_C"], Kp["P_hat"], hw, P_xy=Kp["P_xy"], score=score,
                      in1=in1, want_grid=True, want_idx=True)
    P_hat = dev(Kp["P_hat"], cuda)
    try:
        # gather kernel / automatic choice (plane-streaming kernel where the shape qualifies),
        # with and without the transposed table
        # ... and with the score in the reference's (N, n, F) layout or as the transposed view of an
        # (N, F, n) buffer (what TPS_PP produces)
        sc_ref = dev(score, cuda)
        sc_t = None if score is None else dev(np.ascontiguousarray(score.transpose(0, 2, 1)), cuda).transpose(1, 2)
        for kernel, P_hat_t, sc in ((1, None, sc_ref), (0, None, sc_ref), (1, None, sc_t),
                                    (0, ops.transpose_p_hat(P_hat), sc_ref),
                                    (0, ops.transpose_p_hat(P_hat), sc_t)):
            ops.set_warp_tuning(0, 0, kernel, 0)
            out0, out1, grid, idx = ops.warp(dev(in0, cuda), dev(ctrl, cuda), dev(Kp["hat_C"], cuda),
                                             P_hat, hw, P_xy=dev(Kp["P_xy"], cuda),
                                             score=sc, in1=dev(in1, cuda),
                                             want_grid=True, want_idx=True, P_hat_t=P_hat_t)
            assert_biteq(grid, ref["grid"], f"grid (kernel {kernel})")
            assert_biteq(idx, ref["idx"], f"corner indices (kernel {kernel})")
            assert_biteq(out0, ref["out0"], f"out0 (kernel {kernel})")
            assert_biteq(out1, ref["out1"], f"out1 (kernel {kernel})")
    finally:
        ops.set_warp_tuning(0, 0, 0, 0)


@pytest.mark.parametrize("N,C,H,W,perturb", [
    (5, 3, 32, 100, 0.3),       # the image-pair kernel's geometry on the in-place kernel; odd batch: a lone last image
    (9, 1, 32, 100, 0.8),
    (7, 3, 32, 128, 0.1),       # nrtr_tps++.py:28-33; two quadrant pixels per thread
    (4, 1, 32, 128, 0.9),       # grids that leave the image: out-of-image taps of every kind
    (3, 3, 48, 160, 0.2),       # one image per workgroup, three quadrant pixels per thread
    (5, 1, 48, 160, 0.7),
    (6, 3, 32, 64, 0.3),
    (3, 1, 32, 64, 1.5),
])
def test_inplace_kernel_vs_oracle(cuda, oracle, N, C, H, W, perturb):
    """kernel_choice 6: results staged in place of the consumed planes (tpspp_warp_img.h), every instantiated geometry;
    specials in the image (an out-of-image tap must read as zero, never as 0 * value)."""
    F = 20
    Kc = oracle.classic_constants(F, (H, W))
    ctrl = oracle.classic_initial_ctrl(F)[None] + perturb * synth.dyadic((N, F, 2), "ip.ctrl", N)
    img = synth.dyadic((N, C, H, W), "ip.img", N + 1)
    img.reshape(-1)[::97] = -0.0
    img.reshape(-1)[5::409] = np.inf
    img.reshape(-1)[6::503] = -np.inf
    img.reshape(-1)[7::301] = np.nan
    img.reshape(-1)[11::211] = 1e-42
    ref = oracle.warp(img, ctrl, Kc["inv_delta_C"], Kc["P_hat"], (H, W), want_grid=True, want_idx=True)
    P_hat = dev(Kc["P_hat"], cuda)
    assert ops.table_mirror_symmetry(Kc["P_hat"], (H, W), F) == 1
    prep, packed = ops.prepare_mirror_table(P_hat, (H, W))
    assert packed == ops.TABLE_PACKED | ops.TABLE_SPAN
    try:
        ops.set_warp_tuning(0, 0, 6, 0)
        for want in (True, False):
            out, _, grid, idx = ops.warp(dev(img, cuda), dev(ctrl, cuda), dev(Kc["inv_delta_C"], cuda), P_hat, (H, W),
                                         want_grid=want, want_idx=want, P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)
            if want:
                assert_biteq(grid, ref["grid"], "grid")
                assert_biteq(idx, ref["idx"], "corner indices")
            assert_biteq(out, ref["out0"], "warped")
        with pytest.raises(_lib.TpsppError):        # a geometry it is not instantiated for is refused, not mis-run
            z = lambda *s_: torch.zeros(*s_, device=cuda)      # noqa: E731
            p64 = z(64 * 64, 23)
            t64, f64 = ops.prepare_mirror_table(p64, (64, 64))
            ops.warp(z(2, 3, 64, 64), z(2, 20, 2), z(23, 23), p64, (64, 64), P_hat_t=t64, table_flags=ops.TABLE_MIRROR4 | f64)
    finally:
        ops.set_warp_tuning(0, 0, 0, 0)


@pytest.mark.parametrize("C", [1, 3])
def test_odd_full_batch_pair_and_inplace_kernels(cuda, oracle, C):
    """Odd N under a full-chip launch (256 workgroups, the last one with a lone image): the image-pair kernel and the
    in-place kernel against the LDS-staged kernel, bit for bit, repeatedly.  (Round-2 review: the lone image's flag was
    raised before its DMA had landed; small odd batches passed only because the chip was idle.)"""
    N, F, H, W = 511, 20, 32, 100
    Kc = oracle.classic_constants(F, (H, W))
    inv, P_hat = dev(Kc["inv_delta_C"], cuda), dev(Kc["P_hat"], cuda)
    g = torch.Generator(device=cuda).manual_seed(5)
    img = torch.rand((N, C, H, W), generator=g, device=cuda) * 2 - 1
    ctrl = dev(oracle.classic_initial_ctrl(F), cuda)[None] + 0.1 * (torch.rand((N, F, 2), generator=g, device=cuda) * 2 - 1)
    prep, packed = ops.prepare_mirror_table(P_hat, (H, W))
    try:
        ops.set_warp_tuning(0, 0, 2, 0)
        ref = ops.warp(img, ctrl, inv, P_hat, (H, W), P_hat_t=ops.transpose_p_hat(P_hat), table_flags=ops.TABLE_MIRROR4)[0]
        sel = [0, 255, 509, 510]
        want = oracle.warp(img[sel].cpu().numpy(), ctrl[sel].cpu().numpy(), Kc["inv_delta_C"], Kc["P_hat"], (H, W))["out0"]
        assert_biteq(ref[sel], want, "LDS-staged kernel vs oracle")
        for kernel in (5, 6):
            ops.set_warp_tuning(0, 0, kernel, 0)
            out = torch.empty_like(ref)
            for rep in range(25):
                out.fill_(float("nan"))
                ops.warp(img, ctrl, inv, P_hat, (H, W), out0=out, P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)
                assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), f"kernel {kernel}, repetition {rep}"
    finally:
        ops.set_warp_tuning(0, 0, 0, 0)


def test_launch_shape_does_not_change_results(cuda, oracle):
    Kc = oracle.classic_constants(20, (32, 100))
    N = 19
    ctrl = oracle.classic_initial_ctrl(20)[None] + 0.2 * synth.dyadic((N, 20, 2), "l.ctrl")
    img = synth.dyadic((N, 3, 32, 100), "l.img")
    ref = oracle.warp(img, ctrl, Kc["inv_delta_C"], Kc["P_hat"], (32, 100))["out0"]
    args = (dev(img, cuda), dev(ctrl, cuda), dev(Kc["inv_delta_C"], cuda), dev(Kc["P_hat"], cuda),
            (32, 100))
    try:
        for G, tpb in [(1, 64), (2, 128), (3, 192), (8, 256), (16, 256), (32, 64)]:
            ops.set_warp_tuning(G, tpb, 1)
            assert_biteq(ops.warp(*args)[0], ref, f"G={G} tpb={tpb}")
    finally:
        ops.set_warp_tuning(0, 0, 0)


def test_full_size_batch512_properties(cuda, oracle):
    """BASELINE.json configs[1] at full size: batch 512, 3x32x100, F=20.
    (i) control points on the fiducial lattice (C' = C) => the sampling grid is the pixel-centre
        lattice P itself, up to fp32 rounding of the ill-conditioned sums;
    (ii) sample of images checked bit-for-bit against the oracle;
    (iii) linearity: warp(a*x + y) == a*warp(x) + warp(y) to rounding (same grid)."""
    N = 512
    Kc = oracle.classic_constants(20, (32, 100))
    inv, P_hat = dev(Kc["inv_delta_C"], cuda), dev(Kc["P_hat"], cuda)
    P_hat_t = ops.transpose_p_hat(P_hat)
    img = synth.smooth_image((N, 3, 32, 100), "f.img")
    ident = np.broadcast_to(cases.classic_identity_ctrl(20), (N, 20, 2)).copy()
    out, _, grid, _ = ops.warp(dev(img, cuda), dev(ident, cuda), inv, P_hat, (32, 100),
                               want_grid=True, P_hat_t=P_hat_t, table_flags=ops.TABLE_MIRROR4)
    P = Kc["P"].astype(np.float32)
    assert np.abs(grid.cpu().numpy() - P[None]).max() < 2e-5
    # (the warped image is NOT the input even then: the reference builds P at pixel centres but
    #  samples with align_corners=True, a sub-pixel shift that is part of the behaviour to keep)
    assert np.isfinite(out.cpu().numpy()).all()
    ctrl = ident + 0.05 * synth.dyadic((N, 20, 2), "f.ctrl")
    x, y = synth.dyadic((N, 3, 32, 100), "f.x"), synth.dyadic((N, 3, 32, 100), "f.y")
    w = lambda a: ops.warp(dev(a, cuda), dev(ctrl, cuda), inv, P_hat, (32, 100),
                           P_hat_t=P_hat_t, table_flags=ops.TABLE_MIRROR4)[0].cpu().numpy()
    ox, oy, oz = w(x), w(y), w(0.5 * x + y)
    assert_biteq(ops.warp(dev(x, cuda), dev(ctrl, cuda), inv, P_hat, (32, 100))[0], ox,
                 "gather kernel vs LDS kernel at batch 512")
    assert np.abs(oz - (0.5 * ox + oy)).max() < 1e-5
    sel = np.array([0, 1, 63, 64, 255, 256, 300, 511])
    ref = oracle.warp(x[sel], ctrl[sel], Kc["inv_delta_C"], Kc["P_hat"], (32, 100))["out0"]
    assert_biteq(ox[sel], ref, "batch-512 sample vs oracle")
    # the image-pair kernel (prepared table) at batch 512, with specials in the image: -0.0, +-inf, nan and denormals
    # must come out exactly as the LDS-staged kernel's (out-of-image taps read as zero, never as 0 * value)
    prep, packed = ops.prepare_mirror_table(P_hat, (32, 100))
    xs = x.copy()
    xs.reshape(-1)[::997] = -0.0
    xs.reshape(-1)[5::4099] = np.inf
    xs.reshape(-1)[7::5003] = np.nan
    xs.reshape(-1)[11::3001] = 1e-42
    wide = ident + 0.8 * synth.dyadic((N, 20, 2), "f.wide")       # grids that leave the image: clamped taps
    for c_, x_ in ((ctrl, x), (wide, xs)):
        a = ops.warp(dev(x_, cuda), dev(c_, cuda), inv, P_hat, (32, 100), P_hat_t=P_hat_t,
                     table_flags=ops.TABLE_MIRROR4, want_grid=True, want_idx=True)
        b = ops.warp(dev(x_, cuda), dev(c_, cuda), inv, P_hat, (32, 100), P_hat_t=prep,
                     table_flags=ops.TABLE_MIRROR4 | packed, want_grid=True, want_idx=True)
        c = ops.warp(dev(x_, cuda), dev(c_, cuda), inv, P_hat, (32, 100), P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)
        for k_, what in ((0, "warped"), (2, "grid"), (3, "corner indices")):
            assert (a[k_].cpu().numpy().view(np.uint32) == b[k_].cpu().numpy().view(np.uint32)).all(), \
                f"image-pair kernel vs LDS-staged kernel at batch 512: {what}"
        assert (a[0].cpu().numpy().view(np.uint32) == c[0].cpu().numpy().view(np.uint32)).all()
    # the pre-marshalled call bench.py times (ops.WarpPlan) is the same launch
    out_p = torch.empty((N, 3, 32, 100), device=cuda)
    plan = ops.WarpPlan(dev(x, cuda), dev(ctrl, cuda), inv, P_hat, (32, 100), out_p, P_hat_t=prep,
                        table_flags=ops.TABLE_MIRROR4 | packed)
    plan.run()
    plan.run()
    assert_biteq(out_p, ox, "WarpPlan vs ops.warp")
    # ... and on another stream (tpspp_warp_plan_run_on), ordered by the caller; the plan outlives neither buffer (it holds
    # pointers) and is destroyed with the Python object
    other = torch.cuda.Stream(cuda)
    out_p.zero_()
    other.wait_stream(torch.cuda.current_stream(cuda))
    plan.run(other)
    torch.cuda.current_stream(cuda).wait_stream(other)
    assert_biteq(out_p, ox, "WarpPlan.run(stream) vs ops.warp")
    del plan
    with pytest.raises(ValueError):
        ops.WarpPlan(dev(x, cuda), dev(ctrl, cuda), inv, P_hat, (32, 100), out_p[:, :2])


def test_mirror_symmetry_check_rejects_perturbed_table(oracle):
    Kc = oracle.classic_constants(20, (32, 100))
    assert ops.table_mirror_symmetry(Kc["P_hat"], (32, 100), 20) == 1
    bad = Kc["P_hat"].copy()
    bad[1234, 7] = np.nextafter(bad[1234, 7], np.float32(1.0))
    assert ops.table_mirror_symmetry(bad, (32, 100), 20) == 0


def test_bad_arguments_fail_loudly(cuda):
    from tps_pp_amd import _lib
    with pytest.raises(_lib.TpsppError):
        ops.warp(torch.zeros(1, 1, 4, 4), torch.zeros(1, 4, 2), torch.zeros(7, 7), torch.zeros(16, 7),
                 (4, 4))          # CPU tensors: no fallback
    z = lambda *s: torch.zeros(*s, device=cuda)
    with pytest.raises(_lib.TpsppError):
        ops.warp(z(1, 1, 4, 4), z(1, 70, 2), z(73, 73), z(16, 73), (4, 4))   # F + 3 > 64
    with pytest.raises(ValueError):
        ops.warp(z(1, 1, 4, 4), z(1, 4, 2), z(7, 7), z(15, 7), (4, 4))       # P_hat rows != n


@pytest.mark.parametrize("N,C,H,W", [(3, 3, 32, 160), (2, 1, 64, 256), (3, 4, 48, 160), (5, 3, 64, 200), (4, 3, 32, 100),
                                     (2, 1, 16, 64), (3, 3, 32, 64), (1, 4, 32, 128)])
def test_runtime_geometry_kernel_forced(cuda, oracle, N, C, H, W):
    """`kernel_choice` 7 = REQUIRE the in-place kernel with run-time geometry (an error if the shape does not qualify, so
    a pass proves which kernel ran): bit for bit the oracle, grid and tap indices included, with specials in the image
    (a NaN / inf tap must poison exactly the outputs the oracle's taps reach) -- also for the geometries that normally
    take the instantiated kernels."""
    F = 20
    Kc = oracle.classic_constants(F, (H, W))
    ctrl = oracle.classic_initial_ctrl(F)[None] + 0.4 * synth.dyadic((N, F, 2), "g.ctrl", N)
    img = synth.dyadic((N, C, H, W), "g.img", N + 3).copy()
    flat = img.reshape(-1)
    flat[5::1013] = np.inf
    flat[7::2027] = -np.inf
    flat[11::3001] = np.nan
    flat[13::997] = -0.0
    ref = oracle.warp(img, ctrl, Kc["inv_delta_C"], Kc["P_hat"], (H, W), want_grid=True, want_idx=True)
    P_hat = dev(Kc["P_hat"], cuda)
    prep, packed = ops.prepare_mirror_table(P_hat, (H, W))
    assert packed == ops.TABLE_PACKED | ops.TABLE_SPAN
    ops.set_warp_tuning(kernel_choice=7)
    try:
        for want in (True, False):
            out, _, grid, idx = ops.warp(dev(img, cuda), dev(ctrl, cuda), dev(Kc["inv_delta_C"], cuda), P_hat, (H, W),
                                         want_grid=want, want_idx=want, P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)
            assert_biteq(out, ref["out0"], "warped")
            if want:
                assert_biteq(grid, ref["grid"], "grid")
                assert_biteq(idx, ref["idx"], "corner indices")
    finally:
        ops.set_warp_tuning()


def test_runtime_geometry_kernel_geometry_sweep(cuda, oracle):
    """Every (H, W) of a grid of sizes the kernel's planner accepts (H % 16 == 0, W % 4 == 0, image fits the LDS), C in
    {1, 3, 4}, odd and even batches: forced run-time-geometry kernel, bit for bit the oracle.  Covers every pixel-block
    shape (4 x 8, 8 x 4, 16 x 2, 32 x 1), QP 1-4 and workgroups ("bands") per image 1, 2 and 4."""
    F = 20
    shapes = [(16, 32), (16, 36), (16, 200), (32, 40), (32, 48), (32, 72), (32, 96), (32, 224), (48, 64), (48, 100),
              (64, 64), (64, 128), (64, 160), (96, 128), (128, 64)]
    checked = 0
    for i, (H, W) in enumerate(shapes):
        C = (1, 3, 4)[i % 3]
        if C * H * W * 4 > 150 * 1024:
            C = 1
        N = 2 + (i % 3)
        Kc = oracle.classic_constants(F, (H, W))
        P_hat = dev(Kc["P_hat"], cuda)
        prep, packed = ops.prepare_mirror_table(P_hat, (H, W))
        assert packed == ops.TABLE_PACKED | ops.TABLE_SPAN, (H, W)
        ctrl = oracle.classic_initial_ctrl(F)[None] + 0.3 * synth.dyadic((N, F, 2), "sw.ctrl", i)
        img = synth.dyadic((N, C, H, W), "sw.img", i)
        ref = oracle.warp(img, ctrl, Kc["inv_delta_C"], Kc["P_hat"], (H, W), want_grid=True, want_idx=True)
        ops.set_warp_tuning(kernel_choice=7)
        try:
            out, _, grid, idx = ops.warp(dev(img, cuda), dev(ctrl, cuda), dev(Kc["inv_delta_C"], cuda), P_hat, (H, W),
                                         want_grid=True, want_idx=True, P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)
            out2 = ops.warp(dev(img, cuda), dev(ctrl, cuda), dev(Kc["inv_delta_C"], cuda), P_hat, (H, W), P_hat_t=prep,
                            table_flags=ops.TABLE_MIRROR4 | packed)[0]
        finally:
            ops.set_warp_tuning()
        assert_biteq(grid, ref["grid"], f"grid {H}x{W}")
        assert_biteq(idx, ref["idx"], f"corner indices {H}x{W}")
        assert_biteq(out, ref["out0"], f"warped {H}x{W} C={C}")
        assert_biteq(out2, ref["out0"], f"warped (no optional outputs) {H}x{W} C={C}")
        checked += 1
    assert checked == len(shapes)


SPAN_CASES = [(5, 3, 64, 200), (2, 1, 64, 256), (3, 3, 64, 256), (3, 4, 48, 160), (9, 3, 96, 128), (17, 3, 128, 64),
              (1, 3, 48, 160)]


@pytest.mark.parametrize("N,C,H,W", SPAN_CASES)
def test_span_staging_kernel_forced(cuda, oracle, N, C, H, W):
    """`kernel_choice` 8 = REQUIRE the row-band kernel with span staging (tpspp_warp_span.h; an error if the geometry has no
    third table section, so a pass proves which kernel ran).  Bit for bit the oracle -- grid, tap indices, warped values
    -- for a near-identity transformation (every band's span fits its buffer: the LDS-DMA path), for a violent one (0.45
    noise: bands fold far more rows than the buffer holds and take their taps from global memory, others still stage), with
    every workgroup forced onto the global-memory path (bit 6), with another band count and a small buffer; specials in
    the image (a NaN / inf tap must poison exactly the outputs the oracle's taps reach); batches that are no multiple of
    the 8 XCDs the block -> image mapping interleaves; 64x256x3 does not fit the LDS as a whole image at all.  Round 5,
    late: where four workgroups still fit a CU the kernel requests both regions' WINDOWS (band rows +- 2) at launch instead
    of measuring the spans first, and a wavefront whose taps leave the window takes that mirror pixel from global memory
    (0.45 noise: most wavefronts do); bit 7 of the knob (128) selects the measured-span form -- both forms, every case."""
    F = 20
    Kc = oracle.classic_constants(F, (H, W))
    img = synth.dyadic((N, C, H, W), "sp.img", N + 3).copy()
    flat = img.reshape(-1)
    flat[5::1013] = np.inf
    flat[7::2027] = -np.inf
    flat[11::3001] = np.nan
    flat[13::997] = -0.0
    P_hat = dev(Kc["P_hat"], cuda)
    prep, packed = ops.prepare_mirror_table(P_hat, (H, W))
    assert packed == ops.TABLE_PACKED | ops.TABLE_SPAN
    inv = dev(Kc["inv_delta_C"], cuda)
    try:
        for noise, knob in ((0.03, 0), (0.45, 0), (0.1, 64), (0.1, 2 | (40 << 8)), (0.03, 0 | (150 << 8)), (0.03, 128), (0.45, 128),
                            (0.1, 128 | 2 | (40 << 8))):
            ctrl = oracle.classic_initial_ctrl(F)[None] + noise * synth.dyadic((N, F, 2), "sp.ctrl", N + int(100 * noise))
            ref = oracle.warp(img, ctrl, Kc["inv_delta_C"], Kc["P_hat"], (H, W), want_grid=True, want_idx=True)
            try:
                ops.set_warp_tuning(kernel_choice=8, bands=knob)
                out, _, grid, idx = ops.warp(dev(img, cuda), dev(ctrl, cuda), inv, P_hat, (H, W), want_grid=True, want_idx=True,
                                             P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)
                out2 = ops.warp(dev(img, cuda), dev(ctrl, cuda), inv, P_hat, (H, W), P_hat_t=prep,
                                table_flags=ops.TABLE_MIRROR4 | packed)[0]
            except Exception as e:                       # a band count / budget the planner rejects for this geometry
                if knob & 63 and "do not qualify" in str(e):
                    continue
                raise
            tag = f"{H}x{W} C={C} noise {noise} knob {knob}"
            assert_biteq(grid, ref["grid"], "grid " + tag)
            assert_biteq(idx, ref["idx"], "corner indices " + tag)
            assert_biteq(out, ref["out0"], "warped " + tag)
            assert_biteq(out2, ref["out0"], "warped (no optional outputs) " + tag)
    finally:
        ops.set_warp_tuning()


def test_span_staging_kernel_is_the_default_for_large_geometries(cuda, oracle):
    """Without any tuning a 64x200 / 64x256 image takes the span-staging kernel (the banded in-place kernel staged the whole
    image per band; 64x256x3 fits no LDS and went to the gather kernel) and stays bit for bit the oracle; the banded form is
    still reachable (kernel_choice 9) and gives the same bits; a geometry without the third table section is refused by
    kernel_choice 8."""
    F = 20
    for (N, C, H, W) in ((6, 3, 64, 200), (3, 3, 64, 256)):
        Kc = oracle.classic_constants(F, (H, W))
        P_hat = dev(Kc["P_hat"], cuda)
        prep, packed = ops.prepare_mirror_table(P_hat, (H, W))
        ctrl = oracle.classic_initial_ctrl(F)[None] + 0.2 * synth.dyadic((N, F, 2), "spd.ctrl", H)
        img = synth.dyadic((N, C, H, W), "spd.img", W)
        ref = oracle.warp(img, ctrl, Kc["inv_delta_C"], Kc["P_hat"], (H, W))
        args = (dev(img, cuda), dev(ctrl, cuda), dev(Kc["inv_delta_C"], cuda), P_hat, (H, W))
        kw = dict(P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)
        assert_biteq(ops.warp(*args, **kw)[0], ref["out0"], f"default {H}x{W}")
        # a caller that does not vouch for the three-section layout (no TABLE_SPAN: a buffer sized by an earlier build) never
        # reaches the kernel that reads the third section -- same bits from the other kernels, kernel_choice 8 refused
        kw2 = dict(P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | ops.TABLE_PACKED)
        assert_biteq(ops.warp(*args, **kw2)[0], ref["out0"], f"two-section promise {H}x{W}")
        try:
            ops.set_warp_tuning(kernel_choice=8)
            with pytest.raises(Exception, match="do not qualify"):
                ops.warp(*args, **kw2)
        finally:
            ops.set_warp_tuning()
        try:
            ops.set_warp_tuning(kernel_choice=7)
            assert_biteq(ops.warp(*args, **kw)[0], ref["out0"], f"kernel_choice 7 {H}x{W}")
            if C * H * W * 4 <= 150 * 1024:
                ops.set_warp_tuning(kernel_choice=9)
                assert_biteq(ops.warp(*args, **kw)[0], ref["out0"], f"kernel_choice 9 {H}x{W}")
        finally:
            ops.set_warp_tuning()
    Kc = oracle.classic_constants(F, (32, 100))
    P_hat = dev(Kc["P_hat"], cuda)
    prep, packed = ops.prepare_mirror_table(P_hat, (32, 100))
    try:
        ops.set_warp_tuning(kernel_choice=8)
        with pytest.raises(Exception, match="do not qualify"):
            ops.warp(torch.zeros((2, 3, 32, 100), device=cuda), dev(oracle.classic_initial_ctrl(F)[None].repeat(2, 0), cuda),
                     dev(Kc["inv_delta_C"], cuda), P_hat, (32, 100), P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)
    finally:
        ops.set_warp_tuning()
