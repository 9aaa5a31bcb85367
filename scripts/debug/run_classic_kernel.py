"""Runs the classic warp of one geometry with a forced kernel a few dozen times (for rocprofv3 --pmc passes):
python scripts/debug/run_classic_kernel.py HxW kernel_choice [bands] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPSPreprocessor, constants, ops  # noqa: E402

hw = tuple(int(v) for v in sys.argv[1].split("x"))
kern = int(sys.argv[2])
bands = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device("cuda:0")
mod = TPSPreprocessor(num_fiducial=20, img_size=hw, rectified_img_size=hw, num_img_channel=3).eval().to(dev)
gg = mod.GridGenerator
p_hat_t, flags = gg.prepared_table()
g = torch.Generator(device=dev).manual_seed(99)
ident = torch.from_numpy(constants.classic_identity_ctrl(20)).to(dev)
nbuf = 4
imgs = [torch.rand((512, 3) + hw, generator=g, device=dev) * 2 - 1 for _ in range(nbuf)]
ctrls = [ident[None] + 0.05 * (torch.rand((512, 20, 2), generator=g, device=dev) * 2 - 1) for _ in range(nbuf)]
outs = [torch.empty((512, 3) + hw, device=dev) for _ in range(nbuf)]
ops.set_warp_tuning(0, 0, kern, bands)
plans = [ops.WarpPlan(imgs[j], ctrls[j], gg.inv_delta_C, gg.P_hat, hw, outs[j], P_hat_t=p_hat_t, table_flags=flags) for j in range(nbuf)]
for i in range(reps):
    plans[i % nbuf].run()
torch.cuda.synchronize()
print("done", hw, kern, bands)
