import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import ops, _lib
dev = torch.device("cuda:0")
K = M = Co = 512
x = torch.randn(K, M, device=dev); w = torch.randn(K, Co, device=dev) * 0.05
cw = ops.prep_conv_weight(w.t().reshape(Co, K, 1, 1).contiguous())
xi = x.view(1, K, 1, M)
for _ in range(20): ops.conv2d([xi], cw, 1, False)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 512)(); L.tpspp_debug_sk_trace(buf)
t = np.array(buf[:]).reshape(64, 8)[:16]
names = ["loads issued", "loads landed", "mfma+lds write", "barrier", "end"]
d = t[:, 1:6] - t[:, 0:5]
for i, n in enumerate(names): print(f"{n:16s} median {np.median(d[:, i]):7.0f} cycles (min {d[:, i].min():6.0f} max {d[:, i].max():6.0f})")
print("total", np.median(t[:, 5] - t[:, 0]), "spread of start times over 16 WGs", t[:, 0].max() - t[:, 0].min())
