"""Does issuing independent batches round-robin over several HIP streams raise M1 throughput?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import TPSPreprocessor, constants, ops  # noqa: E402

dev = torch.device("cuda:0")
N, nbuf = 512, 14
mod = TPSPreprocessor(20, (32, 100), (32, 100), 3).eval().to(dev)
gg = mod.GridGenerator
pt, fl = gg.prepared_table()
ident = torch.from_numpy(constants.classic_identity_ctrl(20)).to(dev)
imgs = [torch.rand(N, 3, 32, 100, device=dev) * 2 - 1 for _ in range(nbuf)]
ctrls = [ident[None] + 0.05 * (torch.rand(N, 20, 2, device=dev) * 2 - 1) for _ in range(nbuf)]
outs = [torch.empty(N, 3, 32, 100, device=dev) for _ in range(nbuf)]
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    def run(k):
        for i in range(k):
            j = i % nbuf
            with torch.cuda.stream(streams[i % ns]):
                ops.warp(imgs[j], ctrls[j], gg.inv_delta_C, gg.P_hat, (32, 100), out0=outs[j], P_hat_t=pt, table_flags=fl)
    run(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(3000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{ns} stream(s): {dt / 3000 * 1e6:.2f} us per batch, {3000 * N / dt / 1e6:.1f} M img/s")
