"""One geometry through the forced run-time-geometry kernel (kernel_choice 7) against the oracle, in its own process:
python scripts/debug/run_geo_shape.py HxW C N [kernel_choice]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import tps_oracle as O  # noqa: E402
from tps_pp_amd import ops  # noqa: E402

H, W = (int(v) for v in sys.argv[1].split("x"))
C, N = int(sys.argv[2]), int(sys.argv[3])
kern = int(sys.argv[4]) if len(sys.argv) > 4 else 7
cuda = torch.device("cuda:0")
rng = np.random.default_rng(3)
Kc = O.classic_constants(20, (H, W))
P_hat = torch.from_numpy(Kc["P_hat"]).to(cuda)
prep, packed = ops.prepare_mirror_table(P_hat, (H, W))
ctrl = (O.classic_initial_ctrl(20)[None] + 0.3 * (rng.integers(-64, 64, (N, 20, 2)) / 256.0)).astype(np.float32)
img = (rng.integers(-128, 128, (N, C, H, W)) / 64.0).astype(np.float32)
ref = O.warp(img, ctrl, Kc["inv_delta_C"], Kc["P_hat"], (H, W))
ops.set_warp_tuning(kernel_choice=kern)
out = ops.warp(torch.from_numpy(img).to(cuda), torch.from_numpy(ctrl).to(cuda), torch.from_numpy(Kc["inv_delta_C"]).to(cuda), P_hat,
               (H, W), P_hat_t=prep, table_flags=ops.TABLE_MIRROR4 | packed)[0]
torch.cuda.synchronize()
print(sys.argv[1], "C", C, "N", N, "max err", float(np.abs(out.cpu().numpy() - ref["out0"]).max()))
