"""Run-to-run determinism of the persistent decoder step, and agreement with the launch pipeline.
python scripts/debug/persist_determinism.py [n] [seq] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd.nrtr_head import NRTRDecoder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 37
seq = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
torch.manual_seed(5)
dec = NRTRDecoder(num_classes=93, max_seq_len=seq, start_idx=91, padding_idx=92).eval().to(dev)
with torch.no_grad():
    dec.classifier.weight.mul_(6.0)
enc = torch.randn(n, 64, 512, device=dev)
metas = [dict(valid_ratio=(1.0, 0.7, 0.4)[i % 3]) for i in range(n)]
for cd in ("bf16x3", torch.bfloat16):
    dec.compute_dtype = cd
    with torch.no_grad():
        os.environ["TPSPP_HEAD_NO_PERSIST"] = "1"
        want = dec(None, enc, None, metas, train_mode=False)
        want2 = dec(None, enc, None, metas, train_mode=False)
        del os.environ["TPSPP_HEAD_NO_PERSIST"]
        print(cd, "launch pipeline run-to-run max diff", float((want - want2).abs().max()))
        first = None
        for r in range(reps):
            got = dec(None, enc, None, metas, train_mode=False)
            if first is None:
                first = got
            d1 = (got - first).abs().amax(dim=(1, 2))
            d2 = (got - want).abs().amax(dim=(1, 2))
            bad1 = (d1 > 0).nonzero().flatten().tolist()
            bad2 = (d2 > 1e-4).nonzero().flatten().tolist()
            if bad1 or bad2 or r == 0:
                steps = [(int(b), (got[b] - want[b]).abs().amax(dim=1).gt(1e-4).nonzero().flatten().tolist()[:3]) for b in bad2[:6]]
                print(f"  rep {r}: differs from the first persistent run at images {bad1[:12]}; from the launch pipeline (>1e-4) at "
                      f"{bad2[:12]} first steps {steps}; max vs launch {float(d2.max()):.2e}")
