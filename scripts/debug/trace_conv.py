"""Phase timeline of workgroup 0 of the persistent bf16 convolution (needs a library built with -DTPSPP_CONV_TRACE:
put `#define TPSPP_CONV_TRACE 1` at the top of csrc/tpspp_conv_bf16_persist.hip).  MFMA wavefronts: pairs (wait for a
chunk: start, end), then (epilogue start, end) per tile; loaders: (wait drained: start, end), issued, landed per chunk."""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.getcwd())
from tps_pp_amd import ops, _lib
dev = torch.device("cuda:0"); N = 512
B = lambda c, h, w: ops.Blocked.from_nchw(torch.randn(N, c, h, w, device=dev))
which = sys.argv[1] if len(sys.argv) > 1 else "s1"
cin = 192 if which == "enc0" else 64
cw = ops.prep_conv_weight_bf16(torch.randn(64, cin, 3, 3, device=dev) * 0.05, conv_bias=torch.randn(64, device=dev))
srcs, stride = {"s1": ([B(64, 16, 64)], 1), "s2": ([B(64, 32, 128)], 2), "enc0": ([B(64, 16, 64)] * 3, 1)}[which]
cd = ctypes.CDLL(os.path.join(os.getcwd(), "tps_pp_amd", "libtpspp_hip.so"))
for _ in range(3): ops.conv2d_bf16(srcs, cw, stride, out_blocked=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); ops.conv2d_bf16(srcs, cw, stride, out_blocked=True); b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3
buf = (ctypes.c_longlong * 4096)()
cd.tpspp_debug_conv_trace(buf, 4096)
t = np.array(buf[:]).reshape(16, 256)
t0 = t[0, 0]; span = max(int(r[r > 0].max()) for r in t if (r > 0).any()) - t0
print(f"{which}: launch {us:.1f} us (event), workgroup 0 spans {span} ticks -> {us / span * 1e3:.3f} ns per tick")
for wv in (0, 4, 8, 12):
    r = t[wv]; r = r[r > 0]
    print(f"wavefront {wv}: n={len(r)}", " ".join(f"{(x - t0) * us / span:.2f}" for x in r[:90]))
