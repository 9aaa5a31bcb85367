"""Writes the classic TPS constants (F = 20, 32x100) the kernel lab reads: inv_delta_C, P_hat, identity control points."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import constants  # noqa: E402

K = constants.classic(20, (32, 100))
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "warp_lab_consts.bin"), "wb") as f:
    f.write(np.ascontiguousarray(K["inv_delta_C"], dtype=np.float32).tobytes())
    f.write(np.ascontiguousarray(K["P_hat"], dtype=np.float32).tobytes())
    f.write(constants.classic_identity_ctrl(20).astype(np.float32).tobytes())
