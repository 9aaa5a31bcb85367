"""Every convolution call of the exact-fp32 backbone + TPS++ (batch 512) with its shape, device time (each call timed alone, 10
repetitions back to back), the fp32 matrix rate it reaches (peak 157 TFLOP/s) and its algorithmic bytes at 5 TB/s.
python scripts/debug/backbone_layers_f32.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P
from tps_pp_amd import ops
dev = torch.device("cuda:0"); N = 512
torch.manual_seed(0)
bb = P.build_backbone(dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3], strides=[2, 1, 2, 1, 2])).eval().to(dev)
tps = P.TPS_PP(variant="ResNet45").eval().to(dev)
img = torch.rand(N, 3, 32, 128, device=dev) * 2 - 1
orig = ops.conv2d
rows = []
def wrapped(srcs, cw, stride=(1, 1), relu=True, residual=None, res_mode=0, out=None):
    f = lambda: orig(srcs, cw, stride, relu, residual, res_mode, out)
    o = f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    ts = [(s[0] if isinstance(s, tuple) else s) for s in srcs]
    cin = sum(t.shape[1] for t in ts)
    k = cw.kernel
    flops = 2.0 * o.numel() * cin * k * k
    byts = 4.0 * (sum(t.numel() for t in ts) + o.numel() + (residual.numel() if residual is not None else 0))
    rows.append((cin, o.shape[1], k, stride if isinstance(stride, tuple) else (stride, stride), tuple(ts[0].shape[2:]), tuple(o.shape[2:]), len(ts), res_mode, us,
                 flops / us / 1e6, flops / 157e12 * 1e6, byts / 5e12 * 1e6))
    return o
ops.conv2d = wrapped
with torch.no_grad():
    bb(img, tpsnet=tps, test=True)
tot = 0
for r in rows:
    print("Cin %4d Cout %4d k %d stride %s in %s out %s srcs %d res %d : %7.1f us  %6.1f TFLOP/s  (matrix floor %6.1f us, memory at 5 TB/s %6.1f us)" % r)
    tot += r[8]
print(f"{len(rows)} convolution calls, {tot / 1e3:.2f} ms; 1x1: {sum(r[8] for r in rows if r[2] == 1) / 1e3:.2f} ms, 3x3: {sum(r[8] for r in rows if r[2] == 3) / 1e3:.2f} ms; "
      f"sum of max(matrix floor, memory) {sum(max(r[10], r[11]) for r in rows) / 1e3:.2f} ms")
