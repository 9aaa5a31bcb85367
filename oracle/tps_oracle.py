"""CPU oracle for the TPS grid generator + bilinear sampler (python side).

TEST INFRASTRUCTURE, NOT PRODUCT: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / the reported CPU baseline.
``tps_pp_amd`` never imports it.

Parity pin: against outputs of the reference itself (tests/golden/*.npz, generated in the build
container by tests/golden/make_golden.py, replayed by tests/test_oracle_golden.py).  The reference
has no golden vectors of its own for this path.

Two halves:

* constant builders (float64 numpy, cast to fp32 once) restating
  ``GridGenerator._build_C/_build_inv_delta_C/_build_P/_build_P_hat``
  (/root/reference/mmocr/models/textrecog/preprocessor/tps_preprocessor.py:197-268) and
  ``Attention_Enhanced_TPS._build_C/_build_hat_C/_build_P/_build_P_hat``
  (/root/reference/mmocr/models/textrecog/backbones/tps_pp/tps_pp.py:368-465);
* a ctypes binding of ``libtps_oracle.so`` (oracle/tps_oracle.c): T-solve, grid, sampler, fused warp.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile oracle/tps_oracle.c with gcc (idempotent).  Returns the .so path."""
    so = os.path.join(_HERE, "libtps_oracle.so")
    src = os.path.join(_HERE, "tps_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtps_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("TPS_ORACLE_SO")            # (tests: an AddressSanitizer / UBSan build of the same source)
        if not so:
            so = os.path.join(_HERE, "libtps_oracle.so")
            if not os.path.exists(so):
                build()
        L = ctypes.CDLL(so)
        L.tps_oracle_solve_T.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, _f32p]
        L.tps_oracle_solve_T.restype = None
        L.tps_oracle_grid.argtypes = [_f32p, ctypes.c_int, _f32p, _f32p, _f32p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, _f32p]
        L.tps_oracle_grid.restype = None
        L.tps_oracle_grid_sample.argtypes = [_f32p, _f32p] + [ctypes.c_int] * 6 + \
            [_f32p, _i32p, ctypes.c_int]
        L.tps_oracle_grid_sample.restype = None
        L.tps_oracle_warp.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      _f32p, _f32p, _f32p, _f32p, ctypes.c_int, _f32p,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      _f32p, _f32p, _f32p, _i32p]
        L.tps_oracle_warp.restype = ctypes.c_int
        L.tps_oracle_max_threads.restype = ctypes.c_int
        L.tps_oracle_set_threads.argtypes = [ctypes.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------------------------------------
# constants -- classic RARE TPS (tps_preprocessor.py:197-268)
# ------------------------------------------------------------------------------------------------
def _rbf_matrix(C):
    """R[a,b] = rho^2 ln rho with rho = |C[a]-C[b]|, rho := 1 on the diagonal
    (tps_preprocessor.py:215-223 / tps_pp.py:384-391)."""
    F = C.shape[0]
    R = np.zeros((F, F), dtype=float)
    for i in range(F):
        for j in range(i, F):
            r = np.linalg.norm(C[i] - C[j])
            R[i, j] = r
            R[j, i] = r
    np.fill_diagonal(R, 1)
    return (R ** 2) * np.log(R)


def _inv_delta_C(C):
    """inverse of [[1, C, R], [0, C^T], [0, 1^T]] (tps_preprocessor.py:225-236 / tps_pp.py:393-405)."""
    F = C.shape[0]
    R = _rbf_matrix(C)
    delta = np.concatenate([
        np.concatenate([np.ones((F, 1)), C, R], axis=1),
        np.concatenate([np.zeros((2, 3)), np.transpose(C)], axis=1),
        np.concatenate([np.zeros((1, 3)), np.ones((1, F))], axis=1)], axis=0)
    return np.linalg.inv(delta)


def _rbf_P(C, P, eps=1e-6):
    """d^2 ln(d + eps), d = |P[n] - C[k]| (tps_preprocessor.py:257-266 / tps_pp.py:452-463)."""
    diff = P[:, None, :] - C[None, :, :]
    d = np.linalg.norm(diff, ord=2, axis=2, keepdims=False)
    return np.multiply(np.square(d), np.log(d + eps))


def classic_constants(num_fiducial=20, rectified_img_size=(32, 100)):
    """dict(C, P float64; inv_delta_C (F+3,F+3), P_hat (n,F+3) fp32) for ``GridGenerator``."""
    F = num_fiducial
    Hr, Wr = rectified_img_size
    half = int(F / 2)
    x = np.linspace(-1.0, 1.0, half)
    C = np.concatenate([np.stack([x, -1 * np.ones(half)], axis=1),
                        np.stack([x, np.ones(half)], axis=1)], axis=0)
    gx = (np.arange(-Wr, Wr, 2) + 1.0) / Wr
    gy = (np.arange(-Hr, Hr, 2) + 1.0) / Hr
    P = np.stack(np.meshgrid(gx, gy), axis=2).reshape([-1, 2])
    n = P.shape[0]
    P_hat = np.concatenate([np.ones((n, 1)), P, _rbf_P(C, P)], axis=1)
    return dict(C=C, P=P, inv_delta_C=_inv_delta_C(C).astype(np.float32),
                P_hat=P_hat.astype(np.float32))


def classic_initial_ctrl(num_fiducial=20):
    """Bias of LocalizationNetwork.localization_fc2 (tps_preprocessor.py:130-140), (F, 2) fp32."""
    half = int(num_fiducial / 2)
    x = np.linspace(-1.0, 1.0, half)
    top = np.stack([x, np.linspace(0.0, -1.0, num=half)], axis=1)
    bot = np.stack([x, np.linspace(1.0, 0.0, num=half)], axis=1)
    return np.concatenate([top, bot], axis=0).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# constants -- TPS_PP / Attention_Enhanced_TPS (tps_pp.py:368-465)
# ------------------------------------------------------------------------------------------------
def tpspp_constants(rectified_img_size=(16, 64), point_size=(2, 16)):
    """dict(C, P float64; hat_C (F+3,F+3) [= inverse of delta_C], P_hat (n,F), P_xy (n,2) fp32)."""
    py, px = point_size
    Hr, Wr = rectified_img_size
    cx = np.linspace(0.5, px - 0.5, num=int(px)) / px
    cy = np.linspace(0.5, py - 0.5, num=int(py)) / py
    C = np.stack(np.meshgrid(cx, cy), axis=2).reshape([-1, 2])
    gx = np.linspace(0.5, Wr - 0.5, num=int(Wr)) / Wr
    gy = np.linspace(0.5, Hr - 0.5, num=int(Hr)) / Hr
    P = np.stack(np.meshgrid(gx, gy), axis=2).reshape([-1, 2])
    return dict(C=C, P=P, hat_C=_inv_delta_C(C).astype(np.float32),
                P_hat=_rbf_P(C, P).astype(np.float32), P_xy=P.astype(np.float32))


def tpspp_initial_ctrl(point_size=(2, 16)):
    """Bias of TPE.localization_fc2 (tps_pp.py:279-285), (F, 2) fp32."""
    py, px = point_size
    x = np.linspace(0.1, px - 0.1, num=int(px)) / px
    y = np.linspace(0.1, py - 0.1, num=int(py)) / py
    return np.stack(np.meshgrid(x, y), axis=2).reshape(-1, 2).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# arithmetic (C library)
# ------------------------------------------------------------------------------------------------
def solve_T(inv_delta_C, ctrl):
    inv_delta_C, ctrl = _c(inv_delta_C), _c(ctrl)
    N, F, _ = ctrl.shape
    assert inv_delta_C.shape == (F + 3, F + 3)
    T = np.empty((N, F + 3, 2), dtype=np.float32)
    lib().tps_oracle_solve_T(_p(inv_delta_C), _p(ctrl), N, F, _p(T))
    return T


def build_grid(P_hat, T, P_xy=None, score=None):
    """(N, n, 2) sampling grid.  P_hat (n,F+3) when P_xy is None, else (n,F)."""
    P_hat, T, P_xy, score = _c(P_hat), _c(T), _c(P_xy), _c(score)
    N, K, _ = T.shape
    F = K - 3
    n = P_hat.shape[0]
    assert P_hat.shape[1] == (F if P_xy is not None else F + 3)
    if score is not None:
        assert score.shape == (N, n, F)
    grid = np.empty((N, n, 2), dtype=np.float32)
    lib().tps_oracle_grid(_p(P_hat), P_hat.shape[1], _p(P_xy), _p(score), _p(T), N, n, F, _p(grid))
    return grid


def grid_sample(inp, grid, out_hw, weight_form=2, return_idx=False):
    """bilinear / border / align_corners=True.  inp (N,C,H,W); grid (N, Ho*Wo, 2) or (N,Ho,Wo,2)."""
    inp, grid = _c(inp), _c(grid)
    N, C, H, W = inp.shape
    Ho, Wo = out_hw
    grid = grid.reshape(N, Ho * Wo, 2)
    out = np.empty((N, C, Ho, Wo), dtype=np.float32)
    idx = np.empty((N, Ho * Wo, 2), dtype=np.int32) if return_idx else None
    lib().tps_oracle_grid_sample(_p(inp), _p(grid), N, C, H, W, Ho, Wo, _p(out),
                                 None if idx is None else idx.ctypes.data_as(_i32p), weight_form)
    return (out, idx) if return_idx else out


def warp(in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy=None, score=None, in1=None,
         want_grid=False, want_idx=False, out0=None):
    """Fused hot path; returns dict(out0, out1?, grid?, idx?).  `out0`: a preallocated (N, C0, Ho, Wo) float32 array to
    write into (bench.py's CPU baseline: a fresh 19.7 MB array per call is 4,800 first-touch page faults spread over the
    OpenMP threads -- at 128 threads that, not the arithmetic, was the time of a call)."""
    in0, in1, ctrl, score = _c(in0), _c(in1), _c(ctrl), _c(score)
    inv_delta_C, P_hat, P_xy = _c(inv_delta_C), _c(P_hat), _c(P_xy)
    N, C0, H0, W0 = in0.shape
    F = ctrl.shape[1]
    Ho, Wo = out_hw
    n = Ho * Wo
    if out0 is None:
        out0 = np.empty((N, C0, Ho, Wo), dtype=np.float32)
    elif out0.shape != (N, C0, Ho, Wo) or out0.dtype != np.float32 or not out0.flags["C_CONTIGUOUS"]:
        raise ValueError("warp: out0 must be a C-contiguous float32 array of shape (N, C0, Ho, Wo)")
    C1 = H1 = W1 = 0
    out1 = None
    if in1 is not None:
        _, C1, H1, W1 = in1.shape
        out1 = np.empty((N, C1, Ho, Wo), dtype=np.float32)
    grid = np.empty((N, n, 2), dtype=np.float32) if want_grid else None
    idx = np.empty((N, n, 2), dtype=np.int32) if want_idx else None
    rc = lib().tps_oracle_warp(_p(in0), C0, H0, W0, _p(in1), C1, H1, W1, _p(ctrl), _p(score),
                               _p(inv_delta_C), _p(P_hat), P_hat.shape[1], _p(P_xy), N, F, Ho, Wo,
                               _p(out0), _p(out1), _p(grid),
                               None if idx is None else idx.ctypes.data_as(_i32p))
    if rc != 0:
        raise MemoryError("tps_oracle_warp failed")
    return dict(out0=out0, out1=out1, grid=grid, idx=idx)


def set_threads(t):
    lib().tps_oracle_set_threads(int(t))


def max_threads():
    return int(lib().tps_oracle_max_threads())


def warp_backward(g_out0, in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy=None, score=None, in1=None, g_out1=None,
                  chain_grid=False):
    """Gradients of the warp by PyTorch-CPU autograd through the reference's
    own composition: T = bmm(inv_delta_C, [C'; 0]); rows [1, P, rbf * (0.5 score + 1)]; grid = bmm(rows, T);
    F.grid_sample(bilinear, border, align_corners=True) per input
    (backbones/tps_pp/tps_pp.py:467-496,597-615; preprocessor/tps_preprocessor.py:71-83,270-282).
    Returns dict(g_in0, g_in1 | None, g_ctrl, g_score | None) as numpy arrays.

    chain_grid=True: the forward grid is taken from the C oracle's k-ascending FMA chain (what the HIP
    kernels reproduce bit for bit) instead of torch.bmm, whose summation order depends on the BLAS kernel
    picked for the batch size (an ill-conditioned lattice shows 1e-4 differences between the two); the
    sampler's backward is still ATen's, the two matrix products are transposed in float64."""
    import torch
    with torch.enable_grad():
        if chain_grid:
            return _warp_backward_chain(g_out0, in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy, score, in1, g_out1)
        return _warp_backward(g_out0, in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy, score, in1, g_out1)


def _warp_backward_chain(g_out0, in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy, score, in1, g_out1):
    import torch
    import torch.nn.functional as Fn
    tt = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float32)
    N, F = ctrl.shape[0], ctrl.shape[1]
    T = solve_T(inv_delta_C, ctrl)
    grid = build_grid(P_hat, T, P_xy, score)
    gt = tt(grid).reshape(N, out_hw[0], out_hw[1], 2).requires_grad_(True)
    in0_t = tt(in0).requires_grad_(True)
    in1_t = None if in1 is None else tt(in1).requires_grad_(True)
    loss = (Fn.grid_sample(in0_t, gt, padding_mode="border", align_corners=True) * tt(g_out0)).sum()
    if in1_t is not None:
        loss = loss + (Fn.grid_sample(in1_t, gt, padding_mode="border", align_corners=True) * tt(g_out1)).sum()
    loss.backward()
    gg = gt.grad.numpy().reshape(N, -1, 2).astype(np.float64)
    ph = np.asarray(P_hat, dtype=np.float64)
    n = ph.shape[0]
    if P_xy is not None:
        rbf = np.repeat(ph[None], N, 0)
        fac = np.ones_like(rbf) if score is None else (np.asarray(score, np.float64) * 0.5 + 1)
        rows = np.concatenate([np.ones((N, n, 1)), np.repeat(np.asarray(P_xy, np.float64)[None], N, 0), rbf * fac], 2)
    else:
        rows = np.repeat(ph[None], N, 0)
    g_T = np.einsum("bnk,bnx->bkx", rows, gg)
    g_ctrl = np.einsum("kf,bkx->bfx", np.asarray(inv_delta_C, np.float64), g_T)[:, :F]
    g_score = None
    if score is not None:
        g_score = (0.5 * ph[None] * np.einsum("bnx,bkx->bnk", gg, T.astype(np.float64)[:, 3:])).astype(np.float32)
    return dict(g_in0=in0_t.grad.numpy(), g_in1=None if in1_t is None else in1_t.grad.numpy(),
                g_ctrl=g_ctrl.astype(np.float32), g_score=g_score)


def _warp_backward(g_out0, in0, ctrl, inv_delta_C, P_hat, out_hw, P_xy, score, in1, g_out1):
    import torch
    import torch.nn.functional as Fn
    tt = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    in0_t, ctrl_t, score_t, in1_t = tt(in0), tt(ctrl), tt(score), tt(in1)
    for v in (in0_t, ctrl_t, score_t, in1_t):
        if v is not None:
            v.requires_grad_(True)
    N, F = ctrl_t.shape[0], ctrl_t.shape[1]
    inv, ph = tt(inv_delta_C), tt(P_hat)
    n = ph.shape[0]
    if P_xy is not None:
        rbf = ph.unsqueeze(0).repeat(N, 1, 1)
        if score_t is not None:
            rbf = rbf * (score_t * 0.5 + 1)
        rows = torch.cat([torch.ones((N, n, 1)), tt(P_xy).unsqueeze(0).repeat(N, 1, 1), rbf], dim=2)
    else:
        rows = ph.unsqueeze(0).repeat(N, 1, 1)
    cz = torch.cat((ctrl_t, torch.zeros(N, 3, 2)), dim=1)
    grid = torch.bmm(rows, torch.bmm(inv.unsqueeze(0).repeat(N, 1, 1), cz)).reshape(N, out_hw[0], out_hw[1], 2)
    loss = (Fn.grid_sample(in0_t, grid, padding_mode="border", align_corners=True) * tt(g_out0)).sum()
    if in1_t is not None:
        loss = loss + (Fn.grid_sample(in1_t, grid, padding_mode="border", align_corners=True) * tt(g_out1)).sum()
    loss.backward()
    g = lambda v: None if v is None else v.grad.numpy()  # noqa: E731
    return dict(g_in0=g(in0_t), g_in1=g(in1_t), g_ctrl=g(ctrl_t), g_score=g(score_t))
