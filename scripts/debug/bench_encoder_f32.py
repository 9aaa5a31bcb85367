"""Exact-fp32 encoder (full size, batch 512): ms per call.  python scripts/debug/bench_encoder_f32.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd.nrtr_head import NRTREncoder
dev = torch.device("cuda:0"); torch.manual_seed(0)
enc = NRTREncoder().eval().to(dev)
feat = torch.randn(512, 512, 1, 64, device=dev)
with torch.no_grad():
    enc(feat, None); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): enc(feat, None)
    b.record(); torch.cuda.synchronize()
print(f"fp32 encoder batch 512: {a.elapsed_time(b) / 5:.3f} ms")
