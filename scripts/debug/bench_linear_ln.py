"""The decoder's per-step projections alone: LayerNorm-folded skinny GEMMs, K = 512, M = 512 columns."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import ops
dev = torch.device("cuda:0")
K, M = 512, 512
x = torch.randn(K, M, device=dev)
def t(fn, it=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
g, be = torch.randn(K, device=dev), torch.randn(K, device=dev)
for Co in (512, 1536, 256):
    w = torch.randn(K, Co, device=dev) * 0.05
    f = ops.fold_layernorm(g, be, w, torch.randn(Co, device=dev))
    print(f"Cout {Co}: channel-major {t(lambda: ops.linear_ln(x, f)):.1f} us | token-major {t(lambda: ops.linear_ln(x, f, token_major=True)):.1f} us (incl. the output allocation)")
