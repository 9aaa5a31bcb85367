"""Do all pairs of HIP streams overlap the image-pair kernel's launches equally?  K = 20 launches alternating on two streams,
R regions per pair, pairs interleaved; us per launch (median).  python scripts/debug/stream_pairs.py [nstreams]"""
import os, sys, itertools, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPSPreprocessor, ops, constants
dev = torch.device("cuda:0")
F, C, H, W, B = 20, 3, 32, 100, 512
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mod = TPSPreprocessor(F, (H, W), (H, W), C).eval().to(dev); gg = mod.GridGenerator
pt, flags = gg.prepared_table()
nbuf = 14
g = torch.Generator(device=dev).manual_seed(1)
imgs = [torch.rand((B, C, H, W), generator=g, device=dev) * 2 - 1 for _ in range(nbuf)]
ident = torch.from_numpy(constants.classic_identity_ctrl(F)).to(dev)
ctrls = [ident[None] + 0.05 * (torch.rand((B, F, 2), generator=g, device=dev) * 2 - 1) for _ in range(nbuf)]
outs = [torch.empty((B, C, H, W), device=dev) for _ in range(nbuf)]
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(NS - 1)]
plans = [ops.WarpPlan(imgs[j], ctrls[j], gg.inv_delta_C, gg.P_hat, (H, W), outs[j], P_hat_t=pt, table_flags=flags) for j in range(nbuf)]
for i in range(60): plans[i % nbuf].run()
torch.cuda.synchronize()
K, R = 20, 7
def region(sa, sb, first):
    ss = (streams[sa], streams[sb])
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]; e1 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    for i in range(K):
        if i < 2: e0[i].record(ss[i])
        plans[(first + i) % nbuf].run(ss[i % 2])
    for k in range(2): e1[k].record(ss[k])
    torch.cuda.synchronize()
    return max(b.elapsed_time(e) for e in e1 for b in e0) * 1e3 / K
pairs = list(itertools.combinations(range(NS), 2))
res = {p: [] for p in pairs}
first = 0
for r in range(R):
    for p in pairs:
        res[p].append(region(p[0], p[1], first)); first += K
for p in pairs:
    a = sorted(res[p])
    print(f"streams {p}: median {a[len(a) // 2]:.2f} us  min {a[0]:.2f}  max {a[-1]:.2f}")
