"""The backbone + TPS++ alone (batch 512, 3x32x128) in the configuration given as argv[1] (fp32 | bf16x3 | bf16): for a
kernel trace (rocprofv3 --kernel-trace --stats -- python3 scripts/debug/bench_backbone.py bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P
dev = torch.device("cuda:0"); N = 512
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(0)
bb = P.build_backbone(dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3], strides=[2, 1, 2, 1, 2])).eval().to(dev)
tps = P.TPS_PP(variant="ResNet45").eval().to(dev)
cd = {"fp32": None, "bf16x3": "bf16x3", "bf16": torch.bfloat16}[mode]
bb.compute_dtype = cd; tps.compute_dtype = cd if cd == "bf16x3" else None
img = torch.rand(N, 3, 32, 128, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(3): bb(img, tpsnet=tps, test=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): bb(img, tpsnet=tps, test=True)
    b.record(); torch.cuda.synchronize()
print(f"backbone + TPS++ batch {N} {mode}: {a.elapsed_time(b) / 10:.2f} ms")
