import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPS_PP, ops, constants
dev = torch.device("cuda:0"); N = 512
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
m = TPS_PP().eval().to(dev); at = m.atten_tps; P_xy, P_hat_t = at.device_constants(dev)
g = torch.Generator(device=dev).manual_seed(1)
fg = torch.rand((N, 64, 32, 128), generator=g, device=dev); x = torch.rand((N, 64, 16, 64), generator=g, device=dev)
ctrl = torch.from_numpy(constants.tpspp_initial_ctrl((2, 16))).to(dev)[None].repeat(N, 1, 1).contiguous()
ctrl = ctrl + 0.02 * (torch.rand(ctrl.shape, generator=g, device=dev) - 0.5)
score = (torch.rand((N, 32, 1024), generator=g, device=dev) - 0.5).transpose(1, 2)
g0 = torch.rand((N, 64, 16, 64), generator=g, device=dev); g1 = torch.rand((N, 64, 16, 64), generator=g, device=dev)
_, _, grid, _ = ops.warp(fg, ctrl, at.hat_C, at.P_hat, (16, 64), P_xy=P_xy, score=score, in1=x, want_grid=True, P_hat_t=P_hat_t)
for n0, n1 in [(True, True), (False, True), (True, False), (False, False)]:
    t = timeit(lambda: ops.warp_backward(g0, fg, grid, ctrl, at.hat_C, at.P_hat, (16, 64), P_xy=P_xy, score=score, in1=x, g_out1=g1, P_hat_t=P_hat_t, need_in0=n0, need_in1=n1))
    print(f"need_in0={n0} need_in1={n1}: {t*1e3:.0f} us")
