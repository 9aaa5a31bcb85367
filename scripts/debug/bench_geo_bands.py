"""Run-time-geometry in-place kernel (kernel_choice 7): workgroups per image ("bands") forced to 1 / 2 / 4, batch 512."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from tps_pp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for hw in ((32, 160), (32, 128), (48, 160), (64, 200)):
    for bands in (0, 1, 2, 4):
        ops.set_warp_tuning(0, 0, 7, bands)
        try:
            r = bench.classic_warp_extra(dev, hw, 2)
        except Exception as e:
            print(hw, "bands", bands, "--", str(e)[:80])
            continue
        finally:
            ops.set_warp_tuning(0, 0, 0, 0)
        print(hw, "bands", bands, "2 streams", round(r["launch_us"], 2), round(r["frac_of_hbm_peak"], 3), "| one stream",
              round(r["one_stream"]["launch_us"], 2), round(r["one_stream"]["frac_of_hbm_peak"], 3))
