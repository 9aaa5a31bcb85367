import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPS_PP, ops
dev = torch.device("cuda:0"); N = 512
m = TPS_PP().eval().to(dev)
en = torch.randn(N, 64, 2, 16, device=dev)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
print(f"tpe_points {t(lambda: ops.tpe_points(en, m.TPE)):.0f} us (incl. weight views) | cbam {t(lambda: ops.cbam(en, m.MSFA.conv.atten)):.0f} us")
