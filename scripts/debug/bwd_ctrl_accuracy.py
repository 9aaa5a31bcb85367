"""dL/d control points of the classic warp: our kernel and the reference's fp32 autograd (golden G14), both against float64
autograd of the same graph.  Shows whose rounding the 1e-4 tolerance of tests/test_gpu_backward.py absorbs."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
from oracle import tps_oracle as O  # noqa: E402
from tps_pp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
G = cases.load("warp_backward")
inp, gi = cases.g2_inputs(), cases.g14_inputs()
c = O.classic_constants(cases.CL_F, cases.CL_HW)
inv, ph = torch.from_numpy(c["inv_delta_C"]).to(dev), torch.from_numpy(c["P_hat"]).to(dev)
img = torch.from_numpy(inp["img_smooth"]).to(dev).requires_grad_(True)
ctrl = torch.from_numpy(inp["ctrl"]).to(dev).requires_grad_(True)
out = ops.warp_autograd(img, ctrl, inv, ph, cases.CL_HW)
(out * torch.from_numpy(gi["g_out_cl"]).to(dev)).sum().backward()
ours = ctrl.grad.cpu().double().numpy()
gold = G["cl_g_ctrl"].astype(np.float64)
# float64 truth: the reference's graph (tps_preprocessor.py:270-282 + grid_sample) in double
n = ctrl.shape[0]
cd = torch.from_numpy(inp["ctrl"]).double().requires_grad_(True)
invd, phd = torch.from_numpy(c["inv_delta_C"]).double(), torch.from_numpy(c["P_hat"]).double()
T = torch.matmul(invd[None].expand(n, -1, -1), torch.cat((cd, torch.zeros(n, 3, 2, dtype=torch.float64)), 1))
gr = torch.matmul(phd[None].expand(n, -1, -1), T).reshape(n, cases.CL_HW[0], cases.CL_HW[1], 2)
L = (Fn.grid_sample(torch.from_numpy(inp["img_smooth"]).double(), gr, padding_mode="border", align_corners=True) *
     torch.from_numpy(gi["g_out_cl"]).double()).sum()
L.backward()
truth = cd.grad.numpy()
sc = np.abs(truth).max()
print(f"scale {sc:.3f}: ours vs float64 {np.abs(ours - truth).max() / sc:.2e}, reference fp32 (golden) vs float64 "
      f"{np.abs(gold - truth).max() / sc:.2e}, ours vs golden {np.abs(ours - gold).max() / sc:.2e}")
