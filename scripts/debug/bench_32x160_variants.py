"""32x160 classic warp: the instantiated in-place kernel's variants.  NEEDS the lab switch that was in launch_img_geo while
this was measured (images_per_group of set_warp_tuning selecting <IMGS, QP, NLOAD, WPC> = <2,2,3,1> / <1,2,1,2> instead of
<1,2,2,1>); kept as the record of what was compared: 17.9 / 15.1 / 18.7 us per 512 images on one stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from tps_pp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for name, g in (("1 image / workgroup, 12 + 2 wavefronts", 0), ("2 images / workgroup (pair pipeline), 12 + 3", 2),
                ("1 image / workgroup, 2 workgroups per CU (72 registers)", 3)):
    for rep in range(2):
        ops.set_warp_tuning(g, 0, 6, 0)
        try:
            r = bench.classic_warp_extra(dev, (32, 160), 2)
        finally:
            ops.set_warp_tuning(0, 0, 0, 0)
        print(f"{name:58s} 2 streams {r['launch_us']:.2f} us {r['frac_of_hbm_peak']:.3f} | one stream {r['one_stream']['launch_us']:.2f} us "
              f"{r['one_stream']['frac_of_hbm_peak']:.3f} | err {r['max_abs_err_vs_oracle']}")
