"""Lab: the bench's K = 20 launches of the image-pair kernel as (a) stream launches on 1 / 2 streams, (b) ONE hipGraph of K
kernel nodes -- a linear chain, or S independent branches -- replayed once per region.  us per launch, median of R regions.
python scripts/debug/bench_graph.py [K] [R]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tps_pp_amd import TPSPreprocessor, ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dev = torch.device("cuda:0")
F, C, H, W, B = 20, 3, 32, 100, 512
mod = TPSPreprocessor(F, (H, W), (H, W), C).eval().to(dev)
gg = mod.GridGenerator
pt, flags = gg.prepared_table()
nbuf = 14
g = torch.Generator(device=dev).manual_seed(1)
imgs = [torch.rand((B, C, H, W), generator=g, device=dev) * 2 - 1 for _ in range(nbuf)]
from tps_pp_amd import constants
ident = torch.from_numpy(constants.classic_identity_ctrl(F)).to(dev)
ctrls = [ident[None] + 0.05 * (torch.rand((B, F, 2), generator=g, device=dev) * 2 - 1) for _ in range(nbuf)]
outs = [torch.empty((B, C, H, W), device=dev) for _ in range(nbuf)]
ALG = 76960 * B


def plans_on(st):
    with torch.cuda.stream(st):
        return [ops.WarpPlan(imgs[j], ctrls[j], gg.inv_delta_C, gg.P_hat, (H, W), outs[j], P_hat_t=pt, table_flags=flags) for j in range(nbuf)]


main = torch.cuda.current_stream(dev)
side = [torch.cuda.Stream(dev) for _ in range(3)]
streams = [main] + side
pl = [plans_on(s) for s in streams]
for i in range(30):
    pl[0][i % nbuf].run()
torch.cuda.synchronize()
t_end = time.perf_counter() + 0.3
j = 0
while time.perf_counter() < t_end:
    for _ in range(64):
        outs[j % nbuf].copy_(imgs[j % nbuf]); j += 1
    torch.cuda.synchronize()


def region_streams(S, first):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    for k in range(S):
        e0[k].record(streams[k])
    for i in range(K):
        pl[i % S][(first + i) % nbuf].run()
    for k in range(S):
        e1[k].record(streams[k])
    torch.cuda.synchronize()
    return max(b.elapsed_time(e) for e in e1 for b in e0)


def build_graph(S, first):
    cap = torch.cuda.Stream(dev)
    br = [cap] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    bp = [plans_on(s) for s in br]
    gr = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=cap):
        for s in br[1:]:
            s.wait_stream(cap)
        for i in range(K):
            bp[i % S][(first + i) % nbuf].run()
        for s in br[1:]:
            cap.wait_stream(s)
    return gr, bp


def region_graph(gr):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    gr.replay()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def report(name, regs):
    us = sorted(1e3 * r / K for r in regs)
    med = us[(len(us) - 1) // 2]
    print(f"{name:44s} median {med:6.2f} us ({ALG / med / 1e6 / 8000:.3f})  best {us[0]:6.2f}  worst {us[-1]:6.2f}")


graphs = {S: build_graph(S, 5) for S in (1, 2, 3, 4)}
for S, (gr, _) in graphs.items():
    gr.replay()
torch.cuda.synchronize()
for rnd in range(2):
    for S in (1, 2):
        regs = []
        for r in range(R):
            torch.cuda.synchronize()
            regs.append(region_streams(S, 5 + r * K))
        report(f"{S} stream(s), {K} launches", regs)
    for S, (gr, _) in graphs.items():
        regs = []
        for r in range(R):
            torch.cuda.synchronize()
            regs.append(region_graph(gr))
        report(f"hipGraph, {K} nodes in {S} branch(es)", regs)
# the graph's result is the stream launches' result
ref = [o.clone() for o in outs]
for i in range(K):
    pl[0][(5 + i) % nbuf].run()
torch.cuda.synchronize()
graphs[2][0].replay(); torch.cuda.synchronize()
print("graph outputs equal stream outputs:", all(torch.equal(a, b) for a, b in zip(ref, outs)))
