"""Per-phase times of the persistent decoder step (tpspp_head_set_trace): batch 512, median over workgroups and steps.
python scripts/debug/trace_decoder_step.py [bf16|x3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import _lib  # noqa: E402
from tps_pp_amd.nrtr_head import NRTRDecoder  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "x3"
dev = torch.device("cuda:0")
torch.manual_seed(1)
n, L = 512, 40
dec = NRTRDecoder(num_classes=93, max_seq_len=L, start_idx=91, padding_idx=92).eval().to(dev)
dec.compute_dtype = torch.bfloat16 if mode == "bf16" else "bf16x3"
enc = torch.randn(n, 64, 512, device=dev)
with torch.no_grad():
    for _ in range(2):
        dec(None, enc, None, None, train_mode=False)
    wgs = n // 32 * 16
    buf = torch.zeros((L, wgs, 64), dtype=torch.int64, device=dev)
    _lib.lib().tpspp_head_set_trace(buf.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dec(None, enc, None, None, train_mode=False)
    e1.record()
    torch.cuda.synchronize()
    _lib.lib().tpspp_head_set_trace(None)
t = buf.cpu().numpy().astype(np.float64) / 100.0          # us (100 MHz)
names = ["LN+qkv", "self-attn", "x+fc", "LN+q", "cross-attn", "x+fc2", "LN+w1+gelu", "x+w2"]
d = np.diff(t[:, :, :50], axis=2)                          # (L, wgs, 49): phase durations incl. the barrier wait in front
print(f"{mode}: decoder {e0.elapsed_time(e1):.2f} ms (traced run); per step (first stamp -> last) median "
      f"{np.median(t[:, :, 49] - t[:, :, 0]):.1f} us; launch to launch {np.median(np.diff(t[:, 0, 0])):.1f} us")
per = d[:, :, :48].reshape(L, wgs, 6, 8)
for k, nm in enumerate(names):
    x = per[:, :, :, k]
    print(f"  {nm:12s} median {np.median(x):6.2f} us   mean {x.mean():6.2f}   by step 1/20/39: "
          f"{np.median(per[1, :, :, k]):.2f} / {np.median(per[20, :, :, k]):.2f} / {np.median(per[39, :, :, k]):.2f}")
print(f"  classifier   median {np.median(d[:, :, 48]):6.2f} us")
print(f"  sum of layer-phase medians x 6 = {6 * sum(np.median(per[:, :, :, k]) for k in range(8)):.1f} us")

# sub-stamps (shader clock ticks; 100 MHz s_memtime on this chip? printed raw) of layer 2's x+fc and LN+q phases:
# [entry (weights requested), barrier passed, X staged, epilogue stored, stores drained]
raw = buf.cpu().numpy()
for nm, o in (("x+fc ", 50), ("LN+q ", 56)):
    sub = raw[:, :, o:o + 5].astype(np.float64)
    dd = np.diff(sub, axis=2)
    print(f"  {nm} sub-phases (s_memtime ticks): barrier {np.median(dd[:, :, 0]):.0f}  staging {np.median(dd[:, :, 1]):.0f}  "
          f"product+epilogue {np.median(dd[:, :, 2]):.0f}  drain {np.median(dd[:, :, 3]):.0f}   total {np.median(sub[:, :, 4] - sub[:, :, 0]):.0f}")
