"""Whole-module throughput on the GPU box: TPS_PP.forward (regressor + fused warp) and the split."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import TPS_PP, TPSPreprocessor  # noqa: E402

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


ONLY16 = len(sys.argv) > 2 and sys.argv[2] == "bf16only"      # profile runs: only the bf16 configuration
m = TPS_PP().eval().to(dev)
x = torch.rand(N, 64, 16, 64, device=dev)
o0 = torch.rand(N, 32, 32, 128, device=dev)
o1 = torch.rand(N, 32, 32, 128, device=dev)
with torch.no_grad():
    if len(sys.argv) > 2 and sys.argv[2] == "x3only":
        m.compute_dtype = "bf16x3"
        t3 = timeit(lambda: m(x, [o0, o1]), iters=20)
        print(f"TPS_PP batch {N} bf16x3: full {t3:.2f} ms = {N / t3 * 1e3:,.0f} img/s")
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "fp32only":
        t1 = timeit(lambda: m(x, [o0, o1]), iters=20)
        print(f"TPS_PP batch {N} fp32: full {t1:.2f} ms = {N / t1 * 1e3:,.0f} img/s")
        sys.exit(0)
    if ONLY16:
        xb, o0b, o1b = x.to(torch.bfloat16), o0.to(torch.bfloat16), o1.to(torch.bfloat16)
        t16 = timeit(lambda: m(xb, [o0b, o1b]), iters=20)
        print(f"TPS_PP batch {N} bf16: full {t16:.2f} ms = {N / t16 * 1e3:,.0f} img/s")
        sys.exit(0)
    timeit(lambda: m(x, [o0, o1]), iters=3)            # settle allocator / library caches
    t_reg = timeit(lambda: m.regress(x, [o0, o1]))
    t_full = timeit(lambda: m(x, [o0, o1]))
    cp, sc, fg = m.regress(x, [o0, o1])
    fg = fg.contiguous()
    t_warp = timeit(lambda: m.rectify(fg, x, cp, sc))
print(f"TPS_PP batch {N}: full {t_full:.2f} ms = {N / t_full * 1e3:,.0f} img/s | regressor {t_reg:.2f} ms "
      f"({0.82 * N / t_reg:.1f} TFLOP/s) | warp {t_warp * 1e3:.0f} us")
m.compute_dtype = "bf16x3"        # fp32 tensors, three-term bf16 split in the convolutions (within 1e-4 of the reference)
with torch.no_grad():
    timeit(lambda: m(x, [o0, o1]), iters=3)
    t_reg3 = timeit(lambda: m.regress(x, [o0, o1]))
    t_full3 = timeit(lambda: m(x, [o0, o1]))
print(f"TPS_PP batch {N} bf16x3: full {t_full3:.2f} ms = {N / t_full3 * 1e3:,.0f} img/s | regressor {t_reg3:.2f} ms")
m.compute_dtype = None
p = TPSPreprocessor(20, (32, 100), (32, 100), 3).eval().to(dev)
img = torch.rand(N, 3, 32, 100, device=dev)
with torch.no_grad():
    t_full = timeit(lambda: p(img))
    t_loc = timeit(lambda: p.LocalizationNetwork(img))
print(f"TPSPreprocessor batch {N}: full {t_full:.2f} ms = {N / t_full * 1e3:,.0f} img/s | localisation net {t_loc:.2f} ms")
for mode, tag in (("bf16x3", "bf16x3 (<= 1e-4)"), (torch.bfloat16, "bf16")):
    p.LocalizationNetwork.compute_dtype = mode
    with torch.no_grad():
        t_full = timeit(lambda: p(img))
        t_loc = timeit(lambda: p.LocalizationNetwork(img))
    print(f"TPSPreprocessor batch {N}, {tag} localisation convolutions: full {t_full:.2f} ms = {N / t_full * 1e3:,.0f} img/s | "
          f"localisation net {t_loc:.2f} ms")


# bf16 configuration (BASELINE.json configs[2]): bf16 tensors at the module boundary, bf16 MFMA convolutions
xb, o0b, o1b = x.to(torch.bfloat16), o0.to(torch.bfloat16), o1.to(torch.bfloat16)
with torch.no_grad():
    timeit(lambda: m(xb, [o0b, o1b]), iters=3)
    t_reg16 = timeit(lambda: m.regress(xb, [o0b, o1b]))
    t_full16 = timeit(lambda: m(xb, [o0b, o1b]))
print(f"TPS_PP batch {N} bf16: full {t_full16:.2f} ms = {N / t_full16 * 1e3:,.0f} img/s | regressor {t_reg16:.2f} ms "
      f"({0.82 * N / t_reg16:.1f} TFLOP/s)")
del xb, o0b, o1b
