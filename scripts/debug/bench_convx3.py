"""One bf16x3 3x3 64->64 convolution on 16x64 maps (fp32 tensors, batch 512)."""
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
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05; b = torch.randn(64, device=dev)
cw = ops.prep_conv_weight_bf16(w, conv_bias=b, x3=True)
x1 = torch.randn(N, 64, 16, 64, device=dev)
us = t(lambda: ops.conv2d_bf16([x1], cw, 1, out_dtype=torch.float32))
print(f"bf16x3 3x3 s1 16x64 64->64: {us:.0f} us = {N * 1024 * 64 * 64 * 9 * 2 / us / 1e6:.1f} fp32-equivalent TFLOP/s")
