"""The backbone's wide 3x3 layers (bf16, blocked maps, batch 512) on tpspp_conv3_wide.hip against the tiled kernel
(tpspp_conv_set_tuning bit 2): us per layer, TFLOP/s.  python scripts/debug/bench_wide.py [wide|tiled|both]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import ops, _lib
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
N = 512
for C, H, W in ((128, 8, 32), (256, 8, 32), (512, 4, 16)):
    g = torch.Generator(device="cpu").manual_seed(C)
    x = ops.Blocked.from_nchw(torch.randn((N, C, H, W), generator=g).to(dev))
    r = ops.Blocked.from_nchw(torch.randn((N, C, H, W), generator=g).to(dev))
    w = torch.randn((C, C, 3, 3), generator=g) / np.sqrt(C * 9.0)
    cw = ops.prep_conv_weight_bf16(w.to(dev), conv_bias=torch.zeros(C, device=dev))
    flop = 2.0 * C * C * 9 * H * W * N
    for mode in (("wide", 0), ("tiled", 4)):
        if which not in ("both", mode[0]):
            continue
        _lib.lib().tpspp_conv_set_tuning(mode[1])
        f = lambda: ops.conv2d_bf16([(x, 1, 1)], cw, (1, 1), relu=True, residual=r, res_mode=2, out_blocked=True)
        for _ in range(5): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 30
        print(f"{C:4d} -> {C:4d} @{H}x{W} batch {N} {mode[0]:6s}: {us:7.1f} us  {flop / us / 1e6:7.0f} TFLOP/s ({flop / us / 1e6 / 25:.1f} % of 2.5 PF)")
_lib.lib().tpspp_conv_set_tuning(0)
