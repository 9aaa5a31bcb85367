"""The greedy decoder alone (batch 512, 40 steps) in one configuration (argv[1]: fp32 | bf16x3 | bf16), for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 scripts/debug/bench_decoder.py bf16"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P  # noqa: E402,F401
from tps_pp_amd.nrtr_head import NRTRDecoder  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
torch.manual_seed(0)
dec = NRTRDecoder(num_classes=93, max_seq_len=40, start_idx=91, padding_idx=92).eval().to(dev)
dec.compute_dtype = {"fp32": None, "bf16x3": "bf16x3", "bf16": torch.bfloat16}[mode]
out_enc = torch.randn(512, 64, 512, device=dev)
feat = torch.empty(512, 512, 8, 8, device=dev)
with torch.no_grad():
    for _ in range(2):
        dec(feat, out_enc, None, None, train_mode=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dec(feat, out_enc, None, None, train_mode=False)
    e1.record()
    torch.cuda.synchronize()
print(f"decoder {mode} batch 512: {e0.elapsed_time(e1) / 5:.2f} ms")
