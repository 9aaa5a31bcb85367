import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tps_pp_amd import ops
dev = torch.device("cuda:0")
for hw in ((32, 160), (32, 128), (32, 96), (16, 64)):
    for bands in (0, 8):
        for rep in range(2):
            ops.set_warp_tuning(0, 0, 7, bands)
            try:
                r = bench.classic_warp_extra(dev, hw, 2)
            finally:
                ops.set_warp_tuning(0, 0, 0, 0)
            print(hw, "pair form" if bands == 0 else "one image per workgroup", "2 streams", round(r["launch_us"], 2), round(r["frac_of_hbm_peak"], 3), "| one stream",
                  round(r["one_stream"]["launch_us"], 2), round(r["one_stream"]["frac_of_hbm_peak"], 3))
