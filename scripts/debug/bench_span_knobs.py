"""Span-staging kernel (kernel_choice 8) at 64x200 / 64x256 / 48x160, batch 512: workgroups per image x LDS budget
(occupancy) x forced global-memory path.  One stream, rotating buffers (bench.classic_warp_extra)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from tps_pp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
geoms = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(64, 200)]
for hw in geoms:
    for bands in (2, 3, 4, 6, 8, 16):
        for kb in (30, 38, 52, 78):
            ops.set_warp_tuning(0, 0, 8, bands | (kb << 8))
            try:
                r = bench.classic_warp_extra(dev, hw, 2)
                print(hw, "bands", bands, "lds_kb", kb, "one stream", round(r["one_stream"]["launch_us"], 2),
                      round(r["one_stream"]["frac_of_hbm_peak"], 3), "| 2 streams", round(r["launch_us"], 2), "err", r["max_abs_err_vs_oracle"], flush=True)
            except Exception as e:
                print(hw, "bands", bands, "lds_kb", kb, "--", str(e)[:60], flush=True)
            finally:
                ops.set_warp_tuning(0, 0, 0, 0)
