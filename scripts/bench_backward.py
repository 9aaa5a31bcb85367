"""Backward of the fused warp (row F2) at batch 512: time per call and effective bandwidth."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import TPS_PP, TPSPreprocessor, ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if "fixed" in sys.argv[2:]:            # round 3's fixed-point LDS accumulator instead of the fp64 atomics
    ops.set_warp_bwd_accumulator(True)
    print("accumulator: 64-bit fixed point")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# TPS_PP geometry
m = TPS_PP().eval().to(dev)
at = m.atten_tps
P_xy, P_hat_t = at.device_constants(dev)
g = torch.Generator(device=dev).manual_seed(1)
fg = torch.rand((N, 64, 32, 128), generator=g, device=dev)
x = torch.rand((N, 64, 16, 64), generator=g, device=dev)
from tps_pp_amd import constants  # noqa: E402
ctrl = torch.from_numpy(constants.tpspp_initial_ctrl((2, 16))).to(dev)[None].repeat(N, 1, 1).contiguous()
ctrl = ctrl + 0.02 * (torch.rand(ctrl.shape, generator=g, device=dev) - 0.5)
score = (torch.rand((N, 32, 1024), generator=g, device=dev) - 0.5).transpose(1, 2)
g0 = torch.rand((N, 64, 16, 64), generator=g, device=dev)
g1 = torch.rand((N, 64, 16, 64), generator=g, device=dev)
out0, out1, grid, _ = ops.warp(fg, ctrl, at.hat_C, at.P_hat, (16, 64), P_xy=P_xy, score=score, in1=x, want_grid=True,
                               P_hat_t=P_hat_t)
t = timeit(lambda: ops.warp_backward(g0, fg, grid, ctrl, at.hat_C, at.P_hat, (16, 64), P_xy=P_xy, score=score, in1=x,
                                     g_out1=g1, P_hat_t=P_hat_t))
# algorithmic bytes: read g_out0, g_out1, taps of both inputs (<= inputs once), grid, score; write g_in0, g_in1, g_score
byt = 4 * N * (2 * 64 * 1024 + 64 * 32 * 128 + 64 * 16 * 64 + 2 * 1024 + 32 * 1024 + 64 * 32 * 128 + 64 * 16 * 64 + 32 * 1024)
print(f"TPS_PP warp backward batch {N}: {t * 1e3:.0f} us = {byt / t / 1e6:.0f} GB/s algorithmic")
tf = timeit(lambda: ops.warp(fg, ctrl, at.hat_C, at.P_hat, (16, 64), P_xy=P_xy, score=score, in1=x, P_hat_t=P_hat_t))
print(f"   (forward: {tf * 1e3:.0f} us)")
# the same two sampler backwards on PyTorch-ROCm's own kernels (grid gradients only reach `grid`)
import torch.nn.functional as Fn  # noqa: E402
fg_r, x_r = fg.clone().requires_grad_(True), x.clone().requires_grad_(True)
grid_r = grid.view(N, 16, 64, 2).clone().requires_grad_(True)


def lib_bwd():
    o0 = Fn.grid_sample(fg_r, grid_r, padding_mode="border", align_corners=True)
    o1 = Fn.grid_sample(x_r, grid_r, padding_mode="border", align_corners=True)
    torch.autograd.grad([o0, o1], [fg_r, x_r, grid_r], [g0, g1])


def lib_fwd():
    with torch.no_grad():
        Fn.grid_sample(fg_r, grid_r, padding_mode="border", align_corners=True)
        Fn.grid_sample(x_r, grid_r, padding_mode="border", align_corners=True)


tl, tlf = timeit(lib_bwd), timeit(lib_fwd)
print(f"   (PyTorch-ROCm grid_sample x2 forward+backward {tl * 1e3:.0f} us, forward alone {tlf * 1e3:.0f} us)")

p = TPSPreprocessor(20, (32, 100), (32, 100), 3).eval().to(dev)
gg = p.GridGenerator
P_hat_t, flags = gg.prepared_table()
img = torch.rand((N, 3, 32, 100), generator=g, device=dev)
ctrl = torch.from_numpy(constants.classic_initial_ctrl(20)).to(dev)[None].repeat(N, 1, 1).contiguous()
ctrl = ctrl + 0.05 * (torch.rand(ctrl.shape, generator=g, device=dev) - 0.5)
go = torch.rand((N, 3, 32, 100), generator=g, device=dev)
out, _, grid, _ = ops.warp(img, ctrl, gg.inv_delta_C, gg.P_hat, (32, 100), want_grid=True, P_hat_t=P_hat_t, table_flags=flags)
t = timeit(lambda: ops.warp_backward(go, img, grid, ctrl, gg.inv_delta_C, gg.P_hat, (32, 100), P_hat_t=P_hat_t))
byt = 4 * N * (3 * 3200 * 3 + 2 * 3200)
print(f"classic warp backward batch {N}: {t * 1e3:.0f} us = {byt / t / 1e6:.0f} GB/s algorithmic")
