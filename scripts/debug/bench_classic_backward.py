"""Classic (3x32x100, F=20) warp backward: device time per call for several batch sizes (events around 50 calls).
python scripts/debug/bench_classic_backward.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import constants, ops  # noqa: E402
from tps_pp_amd.tps_preprocessor import TPSPreprocessor  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
p = TPSPreprocessor(20, (32, 100), (32, 100), 3).eval().to(dev)
gg = p.GridGenerator
P_hat_t, flags = gg.prepared_table()
for N in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024, 2048]:
    img = torch.rand((N, 3, 32, 100), generator=g, device=dev)
    ctrl = torch.from_numpy(constants.classic_initial_ctrl(20)).to(dev)[None].repeat(N, 1, 1).contiguous()
    ctrl = ctrl + 0.05 * (torch.rand(ctrl.shape, generator=g, device=dev) - 0.5)
    go = torch.rand((N, 3, 32, 100), generator=g, device=dev)
    _, _, grid, _ = ops.warp(img, ctrl, gg.inv_delta_C, gg.P_hat, (32, 100), want_grid=True, P_hat_t=P_hat_t, table_flags=flags)
    f = lambda: ops.warp_backward(go, img, grid, ctrl, gg.inv_delta_C, gg.P_hat, (32, 100), P_hat_t=P_hat_t)  # noqa: E731
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"N={N}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call", flush=True)
