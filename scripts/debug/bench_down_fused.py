"""The fused down0 + down0_1 kernel (tpspp_down_fused.hip) against the two-kernel route (front's blocked feat0 -> 3x3
stride-2 convolution), and the front with / without its feat0 / feat1 stores: batch 512, 32 x 128 maps.
`python scripts/debug/bench_down_fused.py [N]`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPS_PP, ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
m = TPS_PP().eval().to(dev)
fw = ops.FrontWeightsBf16(m)
cw = ops.prep_conv_weight_bf16(m.down0_1.conv.weight, conv_bias=m.down0_1.conv.bias)
o0 = torch.randn(N, 32, 32, 128, device=dev).bfloat16()
o1 = torch.randn(N, 32, 32, 128, device=dev).bfloat16()
x = torch.randn(N, 64, 16, 64, device=dev).bfloat16()
f0 = ops.front_bf16(o0, o1, x, fw, torch.bfloat16, blocked=True)[0]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rep in range(3):
    a = timeit(lambda: ops.down_fused_bf16(o0, fw.w0, fw.b0, cw))
    b = timeit(lambda: ops.conv2d_bf16([f0], cw, 2, out_blocked=True))
    c = timeit(lambda: ops.front_bf16(o0, o1, x, fw, torch.bfloat16, blocked=True, store01=False))
    d = timeit(lambda: ops.front_bf16(o0, o1, x, fw, torch.bfloat16, blocked=True))
    gb = N * (32 * 32 * 128 + 64 * 16 * 64) * 2 / 1e9
    print(f"batch {N}: down_fused {a:6.1f} us ({gb / a * 1e6 / 1e3:.2f} TB/s of its {gb:.2f} GB) | stride-2 conv on feat0 {b:6.1f} us | "
          f"front without feat0/feat1 stores {c:6.1f} us | front {d:6.1f} us")
