"""Every convolution call of the bf16 backbone + TPS++ (batch 512) with its shape, layouts and device time (each call timed alone,
10 repetitions back to back).  python scripts/debug/backbone_layers.py [bf16|bf16x3]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P
from tps_pp_amd import ops
dev = torch.device("cuda:0"); N = 512
torch.manual_seed(0)
bb = P.build_backbone(dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3], strides=[2, 1, 2, 1, 2])).eval().to(dev)
tps = P.TPS_PP(variant="ResNet45").eval().to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
bb.compute_dtype = torch.bfloat16 if mode == "bf16" else "bf16x3"
if mode != "bf16":
    tps.compute_dtype = "bf16x3"
img = torch.rand(N, 3, 32, 128, device=dev) * 2 - 1
orig = ops.conv2d_bf16
rows = []
def lay(t):
    return ("blk32" if isinstance(t, ops.Blocked32) else "blk") if isinstance(t, ops.Blocked) else ("f32" if t.dtype == torch.float32 else "b16")
def wrapped(srcs, cw, stride=(1, 1), relu=True, residual=None, res_mode=0, out_dtype=torch.bfloat16, out_blocked=False):
    f = lambda: orig(srcs, cw, stride, relu, residual, res_mode, out_dtype, out_blocked)
    out = f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    s0 = srcs[0][0] if isinstance(srcs[0], tuple) else srcs[0]
    cin = sum((s[0] if isinstance(s, tuple) else s).shape[1] for s in srcs)
    oshape = out.shape
    rows.append((cin, oshape[1], getattr(cw, "k", getattr(cw, "kh", "?")), stride if isinstance(stride, tuple) else (stride, stride), tuple(s0.shape[2:]), tuple(oshape[2:]),
                 "+".join(lay(s[0] if isinstance(s, tuple) else s) for s in srcs), "blk" if out_blocked else ("f32" if out_dtype == torch.float32 else "b16"),
                 res_mode, a.elapsed_time(b) * 100))
    return out
ops.conv2d_bf16 = wrapped
import tps_pp_amd.resnet_v2_large as R, tps_pp_amd.tps_pp as T
with torch.no_grad():
    bb(img, tpsnet=tps, test=True)
tot = 0
for r in rows:
    print("Cin %4s Cout %4s k %s stride %s in %s out %s src %-12s dst %s res %d : %7.1f us" % r)
    tot += r[-1]
print(f"{len(rows)} convolution calls, {tot / 1e3:.2f} ms")
