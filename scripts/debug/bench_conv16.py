"""One bf16 convolution (batch 512): 3x3 64->64 on 16x64 maps (stride 1) and 32x128 -> 16x64 (stride 2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import ops
dev = torch.device("cuda:0"); N = 512
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
torch.manual_seed(0)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05; b = torch.randn(64, device=dev)
cw = ops.prep_conv_weight_bf16(w, conv_bias=b)
x1 = torch.randn(N, 64, 16, 64, device=dev).bfloat16(); x2 = torch.randn(N, 64, 32, 128, device=dev).bfloat16()
print(f"3x3 s1 16x64: {t(lambda: ops.conv2d_bf16([x1], cw, 1)):.0f} us | 3x3 s2 32x128->16x64: {t(lambda: ops.conv2d_bf16([x2], cw, 2)):.0f} us")
