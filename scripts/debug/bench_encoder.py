"""NRTR encoder alone (batch 512 x 64 tokens x 512 channels), per arithmetic configuration: the token GEMM
(tpspp_tokgemm.hip) carries the bf16 / bf16x3 projections.  `python scripts/debug/bench_encoder.py [N] [iters]`;
TPSPP_HEAD_NO_TOKGEMM=1 routes them through the convolution kernel as before round 4."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P  # noqa: E402
from tps_pp_amd.nrtr_head import NRTREncoder  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
enc = NRTREncoder().eval().to(dev)
feat = torch.randn(N, 512, 8, 8, device=dev)
with torch.no_grad():
    ref = enc(feat, None)
    for tag, cd in (("fp32", None), ("bf16x3", "bf16x3"), ("bf16", torch.bfloat16)):
        enc.compute_dtype = cd
        out = enc(feat, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            enc(feat, None)
        e1.record()
        torch.cuda.synchronize()
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        print(f"encoder {tag:7s} batch {N}: {e0.elapsed_time(e1) / iters:.2f} ms   max |diff| / max |fp32| = {err:.2e}")
