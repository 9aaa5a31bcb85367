"""Launch-shape sweep of the fused warp kernel on the GPU box (tuning aid, not the bench).

    python scripts/sweep_warp.py [--pp]

Times back-to-back launches over a rotating set of buffers larger than the 256 MB Infinity Cache.
"""
import argparse
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import ops, synth, constants  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pp", action="store_true")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N = a.batch
    if not a.pp:
        K = constants.classic(20, (32, 100))
        inv, P_hat = torch.from_numpy(K["inv_delta_C"]).to(dev), torch.from_numpy(K["P_hat"]).to(dev)
        P_hat_t = ops.transpose_p_hat(P_hat)
        hw, F = (32, 100), 20
        bytes_per_img = 3 * 32 * 100 * 4 * 2 + F * 2 * 4
        nbuf = 16
        ident = constants.classic_identity_ctrl(20)
        ins = [torch.from_numpy(synth.dyadic((N, 3, 32, 100), f"s.img{i}")).to(dev) for i in range(nbuf)]
        ctrls = [torch.from_numpy(ident[None] + 0.05 * synth.dyadic((N, 20, 2), f"s.c{i}")).to(dev)
                 for i in range(nbuf)]
        outs = [torch.empty_like(x) for x in ins]

        # pre-marshalled calls (ops.WarpPlan): 3.4 us of host time per launch instead of ~12, so that the sweep sees
        # the device period and not the Python call overhead
        plans = [ops.WarpPlan(ins[j], ctrls[j], inv, P_hat, hw, outs[j], P_hat_t=P_hat_t, table_flags=ops.TABLE_MIRROR4)
                 for j in range(nbuf)]

        def run(i):
            plans[i % nbuf].run()
    else:
        K = constants.tpspp((16, 64), (2, 16))
        inv, P_hat, P_xy = (torch.from_numpy(K[k]).to(dev) for k in ("hat_C", "P_hat", "P_xy"))
        P_hat_t = ops.transpose_p_hat(P_hat)
        hw, F = (16, 64), 32
        bytes_per_img = 1966336
        nbuf = 2
        init = constants.tpspp_initial_ctrl((2, 16))
        g = torch.Generator(device="cpu").manual_seed(0)
        ins0 = [torch.rand((N, 64, 32, 128), generator=g).to(dev) for i in range(nbuf)]
        ins1 = [torch.rand((N, 64, 16, 64), generator=g).to(dev) for i in range(nbuf)]
        scores = [(torch.rand((N, 32, 1024), generator=g) * 2 - 1).to(dev).transpose(1, 2) for i in range(nbuf)]
        if os.environ.get('SCORE_REF_LAYOUT'):
            scores = [s_.contiguous() for s_ in scores]
        ctrls = [torch.from_numpy(init[None] + 0.02 * synth.dyadic((N, 32, 2), f"s.c{i}")).to(dev)
                 for i in range(nbuf)]
        o0 = [torch.empty((N, 64, 16, 64), device=dev) for i in range(nbuf)]
        o1 = [torch.empty((N, 64, 16, 64), device=dev) for i in range(nbuf)]

        def run(i):
            j = i % nbuf
            ops.warp(ins0[j], ctrls[j], inv, P_hat, hw, P_xy=P_xy, score=scores[j], in1=ins1[j],
                     out0=o0[j], out1=o1[j], P_hat_t=P_hat_t)

    print(f"batch {N}  bytes/img {bytes_per_img}  per launch {bytes_per_img * N / 1e6:.1f} MB")
    configs = [(0, 0, 2, 1), (0, 0, 2, 2), (0, 0, 3, 1), (0, 0, 3, 2)] if not a.pp else [(0, 0, 4, 0)]
    configs += [(G, tpb, 1, 0) for G in ((0, 8) if not a.pp else (1, 4)) for tpb in (256,)]
    for G, tpb, kern, bands in configs:
        if True:
            ops.set_warp_tuning(G, tpb, kern, bands)
            for i in range(20):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            print(f"kernel={kern} bands={bands} G={G:2d} tpb={tpb:3d}: {us:8.2f} us/launch  {N / us:8.2f} Mimg/s  "
                  f"{bytes_per_img * N / us / 1e6:6.3f} TB/s  ({bytes_per_img * N / us / 1e6 / 8.0 * 100:.1f}% of 8 TB/s)")
    ops.set_warp_tuning(0, 0, 0, 0)


if __name__ == "__main__":
    main()
