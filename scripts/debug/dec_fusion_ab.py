"""A/B of the decoder's fused launches (round 4): run once per setting of an opt-in switch (environment variable named by
argv[1], e.g. TPSPP_HEAD_QCROSS), save the greedy decoder's scores for a fixed input, compare with the other run's, time.
`python scripts/debug/dec_fusion_ab.py TPSPP_HEAD_QCROSS [N]`  (spawns itself four times, interleaved)"""
import os
import subprocess
import sys

if len(sys.argv) > 2:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import tps_pp_amd as P  # noqa: F401
    from tps_pp_amd.nrtr_head import NRTRDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dec = NRTRDecoder(num_classes=93, max_seq_len=40, start_idx=91, padding_idx=92).eval().to(dev)
    out_enc = torch.randn(512, 64, 512, device=dev)
    feat = torch.empty(512, 512, 8, 8, device=dev)
    res = {}
    for tag, cd in (("fp32", None), ("bf16x3", "bf16x3"), ("bf16", torch.bfloat16)):
        dec.compute_dtype = cd
        with torch.no_grad():
            out = dec(feat, out_enc, None, None, train_mode=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dec(feat, out_enc, None, None, train_mode=False)
            e1.record()
            torch.cuda.synchronize()
        res[tag] = out.cpu()
        print(f"[{sys.argv[2]}] decoder {tag:7s}: {e0.elapsed_time(e1) / 5:6.2f} ms")
    torch.save(res, sys.argv[3])
    sys.exit(0)

var = sys.argv[1]
import torch  # noqa: E402
os.makedirs("/tmp/dec_ab", exist_ok=True)
for mode, val in (("fused", "1"), ("separate", None), ("fused", "1"), ("separate", None)):
    env = dict(os.environ)
    env.pop(var, None)
    if val:
        env[var] = val
    subprocess.run([sys.executable, __file__, var, mode, f"/tmp/dec_ab/{mode}.pt"], env=env, check=True, timeout=300)
a, b = torch.load("/tmp/dec_ab/fused.pt"), torch.load("/tmp/dec_ab/separate.pt")
for k in a:
    print(f"{k}: scores bit-identical = {torch.equal(a[k], b[k])}, max |diff| = {(a[k] - b[k]).abs().max().item():.3e}")
