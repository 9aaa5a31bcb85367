"""front_bf16 alone (batch 512): bf16 and fp32 feat_grid, and the x3 form."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPS_PP, ops
dev = torch.device("cuda:0"); N = 512
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
m = TPS_PP().eval().to(dev)
x = torch.rand(N, 64, 16, 64, device=dev); o0 = torch.rand(N, 32, 32, 128, device=dev); o1 = torch.rand(N, 32, 32, 128, device=dev)
xb, o0b, o1b = x.bfloat16(), o0.bfloat16(), o1.bfloat16()
fw = ops.FrontWeightsBf16(m); fx = ops.FrontWeightsBf16(m, x3=True)
mb = N * (2 * 32 * 4096 * 2 + 64 * 1024 * 2 + 3 * 64 * 4096 * 2 + 64 * 1024 * 2) / 1e6
t16 = t(lambda: ops.front_bf16(o0b, o1b, xb, fw))
print(f"front bf16 {t16:.0f} us ({mb / t16:.2f} TB/s) | fp32 grid {t(lambda: ops.front_bf16(o0b, o1b, xb, fw, torch.float32)):.0f} us | x3 {t(lambda: ops.front_bf16(o0, o1, x, fx)):.0f} us")
