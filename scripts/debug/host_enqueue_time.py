"""Host-side cost of one bench step (ops.warp call from Python) vs the device period: enqueue 200 steps
after a synchronize and read the clock before the device has caught up."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPSPreprocessor, ops  # noqa: E402

dev = torch.device("cuda:0")
mod = TPSPreprocessor(num_fiducial=20, img_size=(32, 100), rectified_img_size=(32, 100), num_img_channel=3).eval().to(dev)
gg = mod.GridGenerator
p_hat_t, flags = gg.prepared_table()
img = torch.rand((512, 3, 32, 100), device=dev)
ctrl = torch.rand((512, 20, 2), device=dev)
out = torch.empty_like(img)


def step():
    ops.warp(img, ctrl, gg.inv_delta_C, gg.P_hat, (32, 100), out0=out, P_hat_t=p_hat_t, table_flags=flags)


plan = ops.WarpPlan(img, ctrl, gg.inv_delta_C, gg.P_hat, (32, 100), out, P_hat_t=p_hat_t, table_flags=flags)
if len(sys.argv) > 1 and sys.argv[1] == "plan":
    step = plan.run          # noqa: F811
for _ in range(50):
    step()
for n in (50, 200, 1000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e6 * (t1 - t0) / n:.2f} us/call, until drained {1e6 * (t2 - t0) / n:.2f} us/call")
