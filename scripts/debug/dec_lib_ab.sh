#!/bin/bash
# A/B of two builds of the library on the greedy decoder (batch 512, 40 steps, the three arithmetic modes), alternating, on the
# GPU box: bash scripts/debug/dec_lib_ab.sh tps_pp_amd/libtpspp_hip.direct   (the variant is copied over the library in the box's
# scratch copy of the repo and the original restored at the end)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
VAR=$1
cp tps_pp_amd/libtpspp_hip.so /tmp/tpspp_orig.so
cat > /tmp/dec_time.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
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
    out.append(f"{mode} {a.elapsed_time(b) / 5:.2f} ms (checksum {p.double().sum().item():.6f})")
print(" | ".join(out))
PY
for rep in 1 2 3; do
  cp /tmp/tpspp_orig.so tps_pp_amd/libtpspp_hip.so; echo "base    : $(python3 /tmp/dec_time.py 2>/dev/null | tail -1)"
  cp "$VAR" tps_pp_amd/libtpspp_hip.so;            echo "variant : $(python3 /tmp/dec_time.py 2>/dev/null | tail -1)"
done
cp /tmp/tpspp_orig.so tps_pp_amd/libtpspp_hip.so
