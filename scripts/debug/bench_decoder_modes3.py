"""Greedy decoder (full size, batch 512, 40 steps) in the three arithmetic modes: ms per decode + a checksum of the scores.
python scripts/debug/bench_decoder_modes3.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd.nrtr_head import NRTRDecoder
dev = torch.device("cuda:0"); torch.manual_seed(1)
dec = NRTRDecoder(num_classes=93, max_seq_len=40, start_idx=91, padding_idx=92).eval().to(dev)
enc = torch.randn(512, 64, 512, device=dev)
out = []
for mode, cd in (("fp32", None), ("bf16x3", "bf16x3"), ("bf16", torch.bfloat16)):
    dec.compute_dtype = cd
    with torch.no_grad():
        p = dec(None, enc, None, None, train_mode=False); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): p = dec(None, enc, None, None, train_mode=False)
        b.record(); torch.cuda.synchronize()
    out.append(f"{mode} {a.elapsed_time(b) / 5:.2f} ms ({p.double().sum().item():.6f})")
print("greedy decoder batch 512: " + " | ".join(out))
