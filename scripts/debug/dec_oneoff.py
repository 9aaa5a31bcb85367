"""Greedy decoder: the one-off part (encoder keys / values of every layer) against the per-step part, from decodes of 40 and 20
steps (batch 512).  python scripts/debug/dec_oneoff.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd.nrtr_head import NRTRDecoder
dev = torch.device("cuda:0")
enc = torch.randn(512, 64, 512, device=dev)
for mode, cd in (("fp32", None), ("bf16x3", "bf16x3"), ("bf16", torch.bfloat16)):
    ts = {}
    for L in (40, 20):
        torch.manual_seed(1)
        dec = NRTRDecoder(num_classes=93, max_seq_len=L, start_idx=91, padding_idx=92).eval().to(dev)
        dec.compute_dtype = cd
        with torch.no_grad():
            dec(None, enc, None, None, train_mode=False); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): dec(None, enc, None, None, train_mode=False)
            b.record(); torch.cuda.synchronize()
        ts[L] = a.elapsed_time(b) / 5
    step = (ts[40] - ts[20]) / 20
    print(f"{mode}: 40 steps {ts[40]:.2f} ms, 20 steps {ts[20]:.2f} ms -> {step * 1e3:.0f} us per step (average over steps 21-40), one-off + first 20 steps' shortfall {ts[40] - 40 * step:.2f} ms")
