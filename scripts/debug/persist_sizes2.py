"""Which images differ between the persistent step and the launch pipeline, over ragged batch sizes (classifier spread x6:
no tie flips), bf16x3 head, 6 steps, 6 runs each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd.nrtr_head import NRTRDecoder  # noqa: E402

dev = torch.device("cuda:0")
seq = int(os.environ.get("SEQ", "6"))
for n in [int(a) for a in sys.argv[1:]]:
    torch.manual_seed(5)
    dec = NRTRDecoder(num_classes=93, max_seq_len=seq, start_idx=91, padding_idx=92).eval().to(dev)
    with torch.no_grad():
        dec.classifier.weight.mul_(6.0)
    enc = torch.randn(n, 64, 512, device=dev)
    dec.compute_dtype = "bf16x3"
    with torch.no_grad():
        os.environ["TPSPP_HEAD_NO_PERSIST"] = "1"
        want = dec(None, enc, None, None, train_mode=False)
        del os.environ["TPSPP_HEAD_NO_PERSIST"]
        worst, first_step = set(), {}
        for r in range(6):
            got = dec(None, enc, None, None, train_mode=False)
            d = (got - want).abs().amax(dim=2)                      # (n, seq)
            for b in (d.amax(dim=1) > 1e-3).nonzero().flatten().tolist():
                worst.add(b)
                first_step[b] = min(first_step.get(b, 99), int((d[b] > 1e-3).nonzero()[0]))
    print(f"N = {n}: differ: {[(b, first_step[b]) for b in sorted(worst)][:20]}", flush=True)
