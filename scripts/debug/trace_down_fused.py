"""Phase timeline of workgroup 0 of the fused down kernel (needs a library built with -DTPSPP_DOWNF_TRACE: put
`#define TPSPP_DOWNF_TRACE 1` at the top of csrc/tpspp_down_fused.hip or add the flag to build.py's EXTRA).
Stamps per wavefront: start, set-up done, then per step: top, [B ready, products done] x 2 segments, produced, ring ready,
products done, stored."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from tps_pp_amd import TPS_PP, ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = TPS_PP().eval().to(dev)
fw = ops.FrontWeightsBf16(m)
cw = ops.prep_conv_weight_bf16(m.down0_1.conv.weight, conv_bias=m.down0_1.conv.bias)
o0 = torch.randn(N, 32, 32, 128, device=dev).bfloat16()
cd = ctypes.CDLL(os.path.join(os.getcwd(), "tps_pp_amd", "libtpspp_hip.so"))
for _ in range(3):
    ops.down_fused_bf16(o0, fw.w0, fw.b0, cw)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
ops.down_fused_bf16(o0, fw.w0, fw.b0, cw)
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3
buf = (ctypes.c_longlong * 256)()
cd.tpspp_debug_downf_trace(buf, 256)
t = np.array(buf[:]).reshape(4, 64)
t0 = t[:, 0].min()
print(f"launch {us:.1f} us (event); stamps in s_memtime ticks")
for wv in range(4):
    r = t[wv]
    r = r[r > 0]
    rel = (r - t0)
    print(f"wavefront {wv}: set-up {rel[1]} ticks; steps (top, b1, mfma1, b2, mfma2, produced, ring ready, products done, stored), ticks since launch:")
    for i in range(2, min(len(rel), 2 + 9 * 6), 9):
        print("   ", " ".join(f"{x:7d}" for x in rel[i:i + 9]), "  deltas", " ".join(f"{x:5d}" for x in np.diff(rel[i:i + 9])))
