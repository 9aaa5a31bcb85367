"""Classic warp at the other instantiated geometries, batch 512: the in-place kernel (kernel_choice 6) against the
round-1 LDS-staged kernel (2), one stream and three (bench.py's protocol: rotating buffers, pre-marshalled calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from tps_pp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
KERNS = [int(a[1:]) for a in sys.argv[1:] if a.startswith("k")] or [1, 2, 6, 7, 8, 9]
BANDS = [int(a[1:]) for a in sys.argv[1:] if a.startswith("b")] or [0]
GEOMS = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(32, 100), (32, 128), (48, 160), (32, 64), (32, 160), (64, 256), (64, 200)]
for hw in GEOMS:
    for kern in KERNS:          # round-1 LDS kernel, instantiated in-place kernel, in-place kernel with run-time geometry
      for bands in (BANDS if kern == 8 else [0]):
        ops.set_warp_tuning(0, 0, kern, bands)
        try:
            r = bench.classic_warp_extra(dev, hw, 3)
        except Exception as e:          # a geometry the forced kernel does not take
            print(hw, "kernel", kern, "--", str(e)[:80])
            continue
        finally:
            ops.set_warp_tuning(0, 0, 0, 0)
        print(hw, "kernel", kern, "bands", bands, "3 streams", round(r["launch_us"], 2), round(r["frac_of_hbm_peak"], 3), "| one stream",
              round(r["one_stream"]["launch_us"], 2), round(r["one_stream"]["frac_of_hbm_peak"], 3), "| err", r["max_abs_err_vs_oracle"])
