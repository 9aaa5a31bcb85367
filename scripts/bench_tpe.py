"""Where does the non-conv part of the TPS_PP regressor spend its time (PyTorch-ROCm eager)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import TPS_PP  # noqa: E402

dev = torch.device("cuda:0")
N = 512


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


m = TPS_PP().eval().to(dev)
en = torch.rand(N, 64, 2, 16, device=dev)
de = torch.rand(N, 64, 16, 64, device=dev)
with torch.no_grad():
    T = m.TPE
    enf = en.flatten(2).transpose(1, 2)
    print("CBAM            %.3f ms" % timeit(lambda: m.MSFA.conv.atten(en)))
    print("DGAB            %.3f ms" % timeit(lambda: T.atten[0](de, enf)))
    from tps_pp_amd import ops
    dw = ops.DgabWeights(T.atten[0])
    print("DGAB (HIP)      %.3f ms" % timeit(lambda: ops.dgab(de, en.view(N, 64, 32), dw)))
    d = T.atten[0]
    print("  norm1         %.3f ms" % timeit(lambda: d.norm1(de)))
    xn = d.norm1(de)
    print("  attn          %.3f ms" % timeit(lambda: d.attn(xn, enf)))
    print("  mlp           %.3f ms" % timeit(lambda: d.mlp(xn)))
    print("loc fc          %.3f ms" % timeit(lambda: T.localization_fc2(T.localization_fc1(enf).view(N, -1))))
    print("get_score       %.3f ms" % timeit(lambda: T.get_score(enf, de)))
    print("TPE total       %.3f ms" % timeit(lambda: T(en, de)))
