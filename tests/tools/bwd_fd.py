import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import cases
from oracle import tps_oracle as O
from tps_pp_amd import ops, constants
cuda = torch.device("cuda:0")
n = 4
g = torch.Generator(device=cuda).manual_seed(5)
c = O.tpspp_constants(cases.PP_HW, cases.PP_POINT)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
inv, ph, pxy = d(c["hat_C"]), d(c["P_hat"]), d(c["P_xy"])
up = lambda t, hw: torch.nn.functional.interpolate(t, size=hw, mode="bilinear", align_corners=True)
fg = up(torch.rand((n, 64, 4, 16), generator=g, device=cuda), (32, 128)).contiguous()
x = up(torch.rand((n, 64, 2, 8), generator=g, device=cuda), (16, 64)).contiguous()
ctrl = d(constants.tpspp_initial_ctrl((2, 16)))[None].repeat(n, 1, 1) + 0.02 * (torch.rand((n, 32, 2), generator=g, device=cuda) - 0.5)
score = 0.5 * (torch.rand((n, 1024, 32), generator=g, device=cuda) - 0.5)
g0 = torch.rand((n, 64, 16, 64), generator=g, device=cuda) - 0.5
g1 = torch.rand((n, 64, 16, 64), generator=g, device=cuda) - 0.5
_, _, grid, _ = ops.warp(fg, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, score=score, in1=x, want_grid=True)
g_fg, g_x, g_ctrl, g_score = ops.warp_backward(g0, fg, grid, ctrl, inv, ph, cases.PP_HW, P_xy=pxy, score=score, in1=x, g_out1=g1)
o = O.warp_backward(g0.cpu().numpy(), fg.cpu().numpy(), ctrl.cpu().numpy(), c["hat_C"], c["P_hat"], cases.PP_HW, P_xy=c["P_xy"], score=score.cpu().numpy(), in1=x.cpu().numpy(), g_out1=g1.cpu().numpy(), chain_grid=True)
print("g_ctrl vs oracle: max err", np.abs(g_ctrl.cpu().numpy() - o["g_ctrl"]).max(), "scale", np.abs(o["g_ctrl"]).max())
print("grid range", float(grid.min()), float(grid.max()))
def loss(ctrl_):
    o0, o1, _, _ = ops.warp(fg, ctrl_, inv, ph, cases.PP_HW, P_xy=pxy, score=score, in1=x)
    return float((o0.double() * g0.double()).sum() + (o1.double() * g1.double()).sum())
d_c = torch.rand(ctrl.shape, generator=g, device=cuda) - 0.5
an = float((g_ctrl.double() * d_c.double()).sum())
for eps in (1e-3, 2e-4, 5e-5, 2e-5, 5e-6):
    fd = (loss(ctrl + eps * d_c) - loss(ctrl - eps * d_c)) / (2 * eps)
    print(f"eps {eps:g}: fd {fd:.2f} analytic {an:.2f} ratio {fd / an:.3f}")
# float64 reference of the same directional derivative (torch CPU double, autograd)
import torch.nn.functional as Fn
with torch.enable_grad():
    cd = ctrl.cpu().double().requires_grad_(True)
    rows = torch.cat([torch.ones(n, 1024, 1, dtype=torch.float64), torch.from_numpy(c["P_xy"]).double()[None].repeat(n, 1, 1),
                      torch.from_numpy(c["P_hat"]).double()[None] * (score.cpu().double() * 0.5 + 1)], 2)
    T = torch.bmm(torch.from_numpy(c["hat_C"]).double()[None].repeat(n, 1, 1), torch.cat((cd, torch.zeros(n, 3, 2, dtype=torch.float64)), 1))
    gr = torch.bmm(rows, T).reshape(n, 16, 64, 2)
    L = (Fn.grid_sample(fg.cpu().double(), gr, padding_mode="border", align_corners=True) * g0.cpu().double()).sum() + \
        (Fn.grid_sample(x.cpu().double(), gr, padding_mode="border", align_corners=True) * g1.cpu().double()).sum()
    L.backward()
print("float64 autograd directional derivative:", float((cd.grad * d_c.cpu().double()).sum()))
