"""DGAB alone (batch 512, 64 channels): the three arithmetic modes, events around 20 calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPS_PP, ops
dev = torch.device("cuda:0")
m = TPS_PP().eval().to(dev)
blk = m.dgab if hasattr(m, "dgab") else [c for c in m.modules() if type(c).__name__ == "DGAB"][0]
N = 512
x = torch.randn(N, 64, 16, 64, device=dev); y = torch.randn(N, 64, 32, device=dev)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
w = ops.DgabWeights(blk); wb = ops.DgabWeightsBf16(blk); w3 = ops.DgabWeightsBf16(blk, x3=True)
print(f"dgab fp32 {t(lambda: ops.dgab(x, y, w)):.0f} us | bf16x3 {t(lambda: ops.dgab_bf16(x, y, w3)):.0f} us | bf16 {t(lambda: ops.dgab_bf16(x, y, wb)):.0f} us (gate + chain)")
