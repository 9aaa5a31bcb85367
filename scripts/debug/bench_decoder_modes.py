"""Greedy decoder (40 steps, batch 512): launch pipeline vs persistent step kernel (one launch per step / one per decode),
with cluster stagger values.
python scripts/debug/bench_decoder_modes.py [stagger_us ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd.nrtr_head import NRTRDecoder  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
n, L = 512, 40
dec = NRTRDecoder(num_classes=93, max_seq_len=L, start_idx=91, padding_idx=92).eval().to(dev)
enc = torch.randn(n, 64, 512, device=dev)


def t(reps=5):
    with torch.no_grad():
        dec(None, enc, None, None, train_mode=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dec(None, enc, None, None, train_mode=False)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for mode, cd in (("fp32", None), ("bf16x3", "bf16x3"), ("bf16", torch.bfloat16)):
    dec.compute_dtype = cd
    os.environ["TPSPP_HEAD_NO_PERSIST"] = "1"
    base = t()
    del os.environ["TPSPP_HEAD_NO_PERSIST"]
    row = [f"launch pipeline {base:.2f} ms"]
    os.environ["TPSPP_HEAD_STEP_LAUNCHES"] = "1"
    row.append(f"one launch per step {t():.2f}")
    del os.environ["TPSPP_HEAD_STEP_LAUNCHES"]
    for st in [int(a) for a in sys.argv[1:]] or [0, 10, 20, 30, 40, 50]:
        os.environ["TPSPP_HEAD_STAGGER_US"] = str(st)
        row.append(f"stagger {st}: {t():.2f}")
    os.environ.pop("TPSPP_HEAD_STAGGER_US", None)
    print(mode, " | ".join(row), flush=True)
