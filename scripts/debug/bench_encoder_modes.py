"""NRTR encoder (full size, batch 512) in the three arithmetic modes: ms per call.  python scripts/debug/bench_encoder_modes.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd.nrtr_head import NRTREncoder
dev = torch.device("cuda:0"); torch.manual_seed(0)
enc = NRTREncoder().eval().to(dev)
feat = torch.randn(512, 512, 1, 64, device=dev)
out = []
for mode, cd in (("fp32", None), ("bf16x3", "bf16x3"), ("bf16", torch.bfloat16)):
    enc.compute_dtype = cd
    with torch.no_grad():
        y = enc(feat, None); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): y = enc(feat, None)
        b.record(); torch.cuda.synchronize()
    out.append(f"{mode} {a.elapsed_time(b) / 5:.3f} ms ({y.double().sum().item():.4f})")
print("encoder batch 512: " + " | ".join(out))
