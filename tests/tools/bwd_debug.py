import os, sys
import numpy as np, torch, torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import tps_oracle as O
from tps_pp_amd import synth, ops
cuda = torch.device("cuda:0")
N, F, hw = 3, 20, (32, 100)
c = O.classic_constants(F, hw)
ctrl = (O.classic_initial_ctrl(F)[None] + np.array([0.0, 0.6, 3.0], np.float32)[:, None, None] * synth.dyadic((N, F, 2), "bwd.ctrl", 1)).astype(np.float32)
img = synth.smooth_image((N, 2, 16, 40), "bwd.img", 1)
g_out = synth.dyadic((N, 2) + hw, "bwd.gout", 1)
want = O.warp_backward(g_out, img, ctrl, c["inv_delta_C"], c["P_hat"], hw)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
it, ct = d(img).requires_grad_(True), d(ctrl).requires_grad_(True)
out = ops.warp_autograd(it, ct, d(c["inv_delta_C"]), d(c["P_hat"]), hw)
(out * d(g_out)).sum().backward()
got = it.grad.cpu().numpy()
for b in range(N):
    e = np.abs(got[b] - want["g_in0"][b])
    print(b, "max err", e.max(), "at", np.unravel_index(e.argmax(), e.shape), "scale", np.abs(want["g_in0"][b]).max())
# grid check
T = O.solve_T(c["inv_delta_C"], ctrl); grid = O.build_grid(c["P_hat"], T)
_, _, ggrid, _ = ops.warp(d(img), d(ctrl), d(c["inv_delta_C"]), d(c["P_hat"]), hw, want_grid=True)
print("grid bit-equal to oracle chain:", np.array_equal(ggrid.cpu().numpy().view(np.uint32), grid.view(np.uint32)))
# torch-GPU autograd on the same grid
it2 = d(img).requires_grad_(True)
o2 = Fn.grid_sample(it2, ggrid.view(N, hw[0], hw[1], 2), padding_mode="border", align_corners=True)
(o2 * d(g_out)).sum().backward()
print("torch GPU autograd vs oracle:", np.abs(it2.grad.cpu().numpy() - want["g_in0"]).max(), " ours vs torch GPU:", np.abs(it2.grad.cpu().numpy() - got).max())
# torch CPU bmm grid vs chain
with torch.no_grad():
    cz = torch.cat((torch.from_numpy(ctrl), torch.zeros(N, 3, 2)), 1)
    tg = torch.bmm(torch.from_numpy(c["P_hat"]).unsqueeze(0).repeat(N, 1, 1), torch.bmm(torch.from_numpy(c["inv_delta_C"]).unsqueeze(0).repeat(N, 1, 1), cz)).numpy()
print("torch bmm grid bit-equal to chain:", np.array_equal(tg.view(np.uint32), grid.view(np.uint32)), np.abs(tg - grid).max())
