"""The big blocked bf16 3x3 layers of TPS_PP (batch 512): persistent LDS-DMA kernel against the tiled kernel
(tpspp_conv_set_tuning bit 1), with the layer's HBM and MFMA floors beside the measured time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import ops, _lib
dev = torch.device("cuda:0"); N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
torch.manual_seed(0)
B = lambda c, h, w: ops.Blocked.from_nchw(torch.randn(N, c, h, w, device=dev))
def weight(cin):
    return ops.prep_conv_weight_bf16(torch.randn(64, cin, 3, 3, device=dev) * 0.05, conv_bias=torch.randn(64, device=dev))
LAYERS = [
    ("down0_1 s2 64->64 32x128->16x64", [B(64, 32, 128)], 64, 2, {}, "blk"),
    ("enc0 192->64 16x64", [B(64, 16, 64), B(64, 16, 64), B(64, 16, 64)], 192, 1, {}, "blk"),
    ("dec2 up2 64->64 + skip 16x64", [(B(64, 8, 32), 2, 2)], 64, 1, {"residual": B(64, 16, 64), "res_mode": 1}, "blk"),
    ("dec3 64->64 16x64 fp32 out", [B(64, 16, 64)], 64, 1, {}, "f32"),
    ("64->64 16x64", [B(64, 16, 64)], 64, 1, {}, "blk"),
]
for name, srcs, cin, stride, kw, out in LAYERS:
    cw = weight(cin)
    okw = {"out_dtype": torch.float32} if out == "f32" else {"out_blocked": True}
    fn = lambda: ops.conv2d_bf16(srcs, cw, stride, **kw, **okw)
    _lib.lib().tpspp_conv_set_tuning(2); t_tiled = t(fn)
    _lib.lib().tpspp_conv_set_tuning(0); t_pers = t(fn)
    in_b = sum((s[0] if isinstance(s, tuple) else s).t.numel() * 2 for s in srcs) + (kw["residual"].t.numel() * 2 if kw else 0)
    out_b = N * 64 * 16 * 64 * (4 if out == "f32" else 2)
    flop = 2.0 * N * 16 * 64 * 64 * cin * 9
    print(f"{name:34s} tiled {t_tiled:6.1f} us | persistent {t_pers:6.1f} us | HBM floor {(in_b + out_b) / 8e6:5.1f} us, "
          f"MFMA floor {flop / 2.5e9:5.1f} us -> {flop / t_pers / 2.5e9 * 100:4.1f} % of the bf16 peak")
