"""Throughput of the fp32 MFMA conv kernel vs PyTorch-ROCm (MIOpen) on the TPS++ shapes, batch 512."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


SHAPES = [
    ("down0 1x1 32->64 @32x128", 32, 32, 128, 64, 1, (1, 1)),
    ("down0_1 3x3 s2 64->64 @32x128", 64, 32, 128, 64, 3, (2, 2)),
    ("down_feat 1x1 192->64 @32x128", 192, 32, 128, 64, 1, (1, 1)),
    ("k_encoder.0 3x3 192->64 @16x64", 192, 16, 64, 64, 3, (1, 1)),
    ("k_decoder.3 3x3 64->64 @16x64", 64, 16, 64, 64, 3, (1, 1)),
    ("k_encoder.1 3x3 s2 64->64 @16x64", 64, 16, 64, 64, 3, (2, 2)),
]
tot_ours = tot_torch = tot_b16 = tot_tb16 = 0.0
for name, cin, h, w, cout, k, st in SHAPES:
    x = torch.rand(N, cin, h, w, device=dev)
    wgt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    cw = ops.prep_conv_weight(wgt, conv_bias=b)
    out = ops.conv2d([x], cw, st, True)
    t_ours = timeit(lambda: ops.conv2d([x], cw, st, True, out=out))
    with torch.no_grad():
        t_torch = timeit(lambda: F.relu(F.conv2d(x, wgt, b, stride=st, padding=(k - 1) // 2)))
    ho, wo = out.shape[2], out.shape[3]
    flop = 2.0 * N * cout * ho * wo * cin * k * k
    # bf16 MFMA kernel (tpspp_conv_bf16.hip): bf16 tensors in and out
    cwb = ops.prep_conv_weight_bf16(wgt, conv_bias=b)
    xb = x.to(torch.bfloat16)
    t_b16 = timeit(lambda: ops.conv2d_bf16([xb], cwb, st, True))
    cw3 = ops.prep_conv_weight_bf16(wgt, conv_bias=b, x3=True)          # bf16x3 on fp32 tensors
    t_x3 = timeit(lambda: ops.conv2d_bf16([x], cw3, st, True, out_dtype=torch.float32))
    with torch.no_grad():
        wb, bb = wgt.to(torch.bfloat16), b.to(torch.bfloat16)
        t_tb16 = timeit(lambda: F.relu(F.conv2d(xb, wb, bb, stride=st, padding=(k - 1) // 2)))
    print(f"{name:36s} fp32 ours {t_ours:7.3f} ms {flop / t_ours / 1e9:6.1f} TF | torch {t_torch:7.3f} ms {flop / t_torch / 1e9:6.1f} TF"
          f" || bf16x3 {t_x3:7.3f} ms {flop / t_x3 / 1e9:6.1f} TF"
          f" || bf16 ours {t_b16:7.3f} ms {flop / t_b16 / 1e9:6.1f} TF | torch {t_tb16:7.3f} ms {flop / t_tb16 / 1e9:6.1f} TF")
    tot_ours += t_ours
    tot_torch += t_torch
    tot_b16 += t_b16
    tot_tb16 += t_tb16
print(f"sum: fp32 ours {tot_ours:.2f} ms, torch {tot_torch:.2f} ms; bf16 ours {tot_b16:.2f} ms, torch {tot_tb16:.2f} ms")
