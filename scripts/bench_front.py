import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import TPS_PP, ops
dev = torch.device("cuda:0"); N = 512
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
m = TPS_PP().eval().to(dev)
x = torch.rand(N, 64, 16, 64, device=dev); o0 = torch.rand(N, 32, 32, 128, device=dev); o1 = torch.rand(N, 32, 32, 128, device=dev)
fw = ops.FrontWeights(m)
print("front fused   %.3f ms" % timeit(lambda: ops.front(o0, o1, x, fw)))
cw = m._conv_weights()
def unfused():
    f0 = ops.conv2d([o0], cw["down0"], 1); f1 = ops.conv2d([o1], cw["down1"], 1); f2 = ops.conv2d([x], cw["down2"], 1)
    return ops.conv2d([f0, f1, (f2, 2, 2)], cw["down_feat"], 1)
print("front unfused %.3f ms" % timeit(unfused))
