"""Phase timeline of the LDS-staged warp kernel (shader-clock stamps per workgroup)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tps_pp_amd import ops, synth, constants, _lib  # noqa: E402

dev = torch.device("cuda:0")
N = 512
K = constants.classic(20, (32, 100))
inv, P_hat = torch.from_numpy(K["inv_delta_C"]).to(dev), torch.from_numpy(K["P_hat"]).to(dev)
P_hat_t = ops.transpose_p_hat(P_hat)
nbuf = 8
ident = constants.classic_identity_ctrl(20)
ins = [torch.from_numpy(synth.dyadic((N, 3, 32, 100), f"s.img{i}")).to(dev) for i in range(nbuf)]
ctrls = [torch.from_numpy(ident[None] + 0.05 * synth.dyadic((N, 20, 2), f"s.c{i}")).to(dev) for i in range(nbuf)]
outs = [torch.empty_like(x) for x in ins]
bands = int(sys.argv[1]) if len(sys.argv) > 1 else 0
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
trace = torch.zeros((2048, 8), dtype=torch.int64, device=dev)
ops.set_warp_tuning(0, 0, 2, bands)
for i in range(10):
    ops.warp(ins[i % nbuf], ctrls[i % nbuf], inv, P_hat, (32, 100), out0=outs[i % nbuf], P_hat_t=P_hat_t, table_flags=FLAGS)
torch.cuda.synchronize()
_lib.lib().tpspp_warp_set_trace(trace.data_ptr())
names = ["start", "T ready", "grid done", "img in LDS", "stores done", "DMA issued", "DMA landed"]
for rep in range(3):
    j = (rep + 3) % nbuf
    ops.warp(ins[j], ctrls[j], inv, P_hat, (32, 100), out0=outs[j], P_hat_t=P_hat_t, table_flags=FLAGS)
    torch.cuda.synchronize()
    t = trace.cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    rel = (t[:, :7] - t0)
    print(f"rep {rep}: s_memtime ticks relative to the earliest workgroup start")
    for i, nm in enumerate(names):
        c = rel[:, i]
        print(f"  {nm:12s} min {c.min():9.0f}  median {np.median(c):9.0f}  max {c.max():9.0f}")
    # s_memtime counts at a constant 2.4 GHz (scripts/ubench/clock_bench) but its origin differs per
    # XCD: spans are only meaningful inside one XCD (workgroup b runs on XCD b % 8)
    nb = t.shape[0]
    for x in range(8):
        sel = t[np.arange(nb) % 8 == x]
        if len(sel) == 0:
            continue
        s0 = sel[:, 0].min()
        print(f"  XCD {x}: {len(sel)} WGs, starts spread {(sel[:, 0].max() - s0) / 2400:.2f} us, "
              f"first start -> last stores retired {(sel[:, 4].max() - s0) / 2400:.2f} us, "
              f"median WG lifetime {np.median(sel[:, 4] - sel[:, 0]) / 2400:.2f} us")
    w = t[:, 7]
    life = (t[:, 4] - t[:, 0]) / 2400.0
    st = (w - w.min()) / 100.0
    print(f"  wall clock: WG starts spread over {st.max():.2f} us (median {np.median(st):.2f}); "
          f"first start -> last WG end {np.max(st + life):.2f} us")
    d = t[:, 1:5] - t[:, 0:4]
    print("  per-WG phase medians: T %.0f | grid %.0f | wait img %.0f | sample+store %.0f" % tuple(np.median(d, axis=0)))
_lib.lib().tpspp_warp_set_trace(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(200):
    ops.warp(ins[i % nbuf], ctrls[i % nbuf], inv, P_hat, (32, 100), out0=outs[i % nbuf], P_hat_t=P_hat_t, table_flags=FLAGS)
e1.record()
torch.cuda.synchronize()
print("us/launch", e0.elapsed_time(e1) * 1e3 / 200)
