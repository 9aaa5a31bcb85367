import os, sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from tps_pp_amd import TPSPreprocessor, ops, constants
dev = torch.device("cuda:0")
F, C, H, W, B = 20, 3, 32, 100, 512
mod = TPSPreprocessor(F, (H, W), (H, W), C).eval().to(dev); gg = mod.GridGenerator
pt, flags = gg.prepared_table()
nbuf = 14
g = torch.Generator(device=dev).manual_seed(1)
imgs = [torch.rand((B, C, H, W), generator=g, device=dev) * 2 - 1 for _ in range(nbuf)]
ident = torch.from_numpy(constants.classic_identity_ctrl(F)).to(dev)
ctrls = [ident[None] + 0.05 * (torch.rand((B, F, 2), generator=g, device=dev) * 2 - 1) for _ in range(nbuf)]
outs = [torch.empty((B, C, H, W), device=dev) for _ in range(nbuf)]
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]
pl = []
for st in streams:
    with torch.cuda.stream(st):
        pl.append([ops.WarpPlan(imgs[j], ctrls[j], gg.inv_delta_C, gg.P_hat, (H, W), outs[j], P_hat_t=pt, table_flags=flags) for j in range(nbuf)])
for i in range(40): pl[0][i % nbuf].run()
torch.cuda.synchronize()
K = 20
for rep in range(8):
    torch.cuda.synchronize()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]; e1 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    for k in range(2): e0[k].record(streams[k])
    t1 = time.perf_counter()
    stamps = []
    for i in range(K):
        pl[i % 2][(rep * K + i) % nbuf].run(); stamps.append(time.perf_counter())
    t2 = time.perf_counter()
    for k in range(2): e1[k].record(streams[k])
    torch.cuda.synchronize()
    gpu = max(b.elapsed_time(e) for e in e1 for b in e0) * 1e3
    d = np.diff([t1] + stamps) * 1e6
    print(f"region {rep}: GPU {gpu / K:.2f} us/launch ({gpu:.0f} us); host enqueue {1e6 * (t2 - t1):.0f} us total, per call median {np.median(d):.1f} max {d.max():.1f} us; events {1e6 * (t1 - t0):.0f} us")
