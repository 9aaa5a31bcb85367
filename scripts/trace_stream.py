"""Phase timeline of the plane-streaming kernel at the TPS_PP geometry (per-workgroup stamps)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import ops, synth, constants, _lib  # noqa: E402

dev = torch.device("cuda:0")
N = 512
K = constants.tpspp((16, 64), (2, 16))
inv, P_hat, P_xy = (torch.from_numpy(K[k]).to(dev) for k in ("hat_C", "P_hat", "P_xy"))
P_hat_t = ops.transpose_p_hat(P_hat)
g = torch.Generator(device="cpu").manual_seed(0)
in0 = torch.rand((N, 64, 32, 128), generator=g).to(dev)
in1 = torch.rand((N, 64, 16, 64), generator=g).to(dev)
score = (torch.rand((N, 32, 1024), generator=g) * 2 - 1).to(dev).transpose(1, 2)
ctrl = torch.from_numpy(constants.tpspp_initial_ctrl((2, 16))[None] + 0.02 * synth.dyadic((N, 32, 2), "c")).to(dev)
o0 = torch.empty((N, 64, 16, 64), device=dev)
o1 = torch.empty((N, 64, 16, 64), device=dev)
trace = torch.zeros((N, 8), dtype=torch.int64, device=dev)


def run():
    ops.warp(in0, ctrl, inv, P_hat, (16, 64), P_xy=P_xy, score=score, in1=in1, out0=o0, out1=o1, P_hat_t=P_hat_t)


for _ in range(3):
    run()
torch.cuda.synchronize()
_lib.lib().tpspp_warp_set_trace(trace.data_ptr())
run()
torch.cuda.synchronize()
_lib.lib().tpspp_warp_set_trace(0)
t = trace.cpu().numpy().astype(np.float64)
w = (t[:, 7] - t[:, 7].min()) / 100.0
life = (t[:, 4] - t[:, 0]) / 2400.0
print(f"WG starts: first wave of {np.sum(w < 5)} WGs within 5 us; last WG starts at {w.max():.1f} us; "
      f"last end {np.max(w + life):.1f} us")
print(f"per-WG (us): T ready {np.median(t[:, 1] - t[:, 0]) / 2400:.2f} | grid+taps {np.median(t[:, 2] - t[:, 1]) / 2400:.2f} | "
      f"stream {np.median(t[:, 4] - t[:, 2]) / 2400:.2f} | lifetime median {np.median(life):.1f} min {life.min():.1f} max {life.max():.1f}")
first = w < 5
print(f"first-wave WGs lifetime {np.median(life[first]):.1f} us, second-wave {np.median(life[~first]):.1f} us")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print("us/launch", e0.elapsed_time(e1) * 100)
