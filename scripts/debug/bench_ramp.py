"""Does the launch period of the bench workload drift with how long the GPU has been busy?  (clock ramp)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPSPreprocessor, constants, ops
dev = torch.device("cuda:0")
B, C, H, W, F = 512, 3, 32, 100, 20
mod = TPSPreprocessor(F, (H, W), (H, W), C).eval().to(dev)
gg = mod.GridGenerator
pt, fl = gg.prepared_table()
nbuf = 14
g = torch.Generator(device=dev).manual_seed(1)
ident = torch.from_numpy(constants.classic_identity_ctrl(F)).to(dev)
imgs = [torch.rand((B, C, H, W), generator=g, device=dev) * 2 - 1 for _ in range(nbuf)]
ctrls = [ident[None] + 0.05 * (torch.rand((B, F, 2), generator=g, device=dev) * 2 - 1) for _ in range(nbuf)]
outs = [torch.empty((B, C, H, W), device=dev) for _ in range(nbuf)]
plans = [ops.WarpPlan(imgs[j], ctrls[j], gg.inv_delta_C, gg.P_hat, (H, W), outs[j], P_hat_t=pt, table_flags=fl) for j in range(nbuf)]
t_start = time.perf_counter()
def run(k):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(k):
        plans[i % nbuf].run()
    t1 = time.perf_counter()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k, (t1 - t0) * 1e6 / k
for k in (20, 20, 200, 2000, 2000, 20000, 20000, 20000, 2000, 20, 20):
    dev_us, host_us = run(k)
    print(f"t={time.perf_counter()-t_start:6.2f}s  {k:6d} launches: {dev_us:6.2f} us/launch (events), host enqueue {host_us:5.2f} us/call")
