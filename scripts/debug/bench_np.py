"""Image-pair kernel with NP pairs per workgroup (tpspp_warp_pair.h): parity with the one-pair form and the launch period under
bench.py's protocols (K launches round-robin on S streams, R regions per variant, variants interleaved region by region).

    python scripts/debug/bench_np.py [K] [R]

The pairs-per-workgroup choice comes from the per-call flag bits (ops.WARP_PAIRS_2 / WARP_PAIRS_4) when the library has them,
else from the lab knob ops.set_warp_tuning(0, 0, 5, np)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P
from tps_pp_amd import ops, constants

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device("cuda:0")
F, H, W = 20, 32, 100
mod = P.TPSPreprocessor(num_fiducial=F, img_size=(H, W), rectified_img_size=(H, W), num_img_channel=1).eval().to(dev)
gg = mod.GridGenerator
p_hat_t, flags = gg.prepared_table()
ident = torch.from_numpy(constants.classic_identity_ctrl(F)).to(dev)
g = torch.Generator(device=dev).manual_seed(7)
HAVE_FLAGS = hasattr(ops, "WARP_PAIRS_2")


def np_flags(npw):
    if not HAVE_FLAGS:
        return 0
    return {1: 0, 2: ops.WARP_PAIRS_2, 4: ops.WARP_PAIRS_4}[npw]


def set_np(npw):
    if not HAVE_FLAGS:
        ops.set_warp_tuning(0, 0, 5 if npw > 1 else 0, npw if npw > 1 else 0)


# ---- parity: every NP against NP = 1, bit for bit, incl. odd and tiny batches, grid / index outputs, 3 channels ----
for C in (1, 3):
    for N in (512, 511, 5, 1, 2, 257, 1024, 1030):
        img = torch.rand((N, C, H, W), generator=g, device=dev) * 2 - 1
        ctrl = ident[None] + 0.08 * (torch.rand((N, F, 2), generator=g, device=dev) * 2 - 1)
        ref = None
        for npw in (1, 2, 4):
            set_np(npw)
            out, _, grid, idx = ops.warp(img, ctrl, gg.inv_delta_C, gg.P_hat, (H, W), P_hat_t=p_hat_t, table_flags=flags | np_flags(npw),
                                         want_grid=True, want_idx=True)
            out2 = ops.warp(img, ctrl, gg.inv_delta_C, gg.P_hat, (H, W), P_hat_t=p_hat_t, table_flags=flags | np_flags(npw))[0]
            torch.cuda.synchronize()
            if ref is None:
                ref = (out.clone(), grid.clone(), idx.clone())
                assert torch.equal(out2, out)
            else:
                ok = torch.equal(out, ref[0]) and torch.equal(grid, ref[1]) and torch.equal(idx, ref[2]) and torch.equal(out2, ref[0])
                print(f"C {C} N {N:5d} np {npw}: {'bit-identical' if ok else 'MISMATCH'}", flush=True)
                assert ok
set_np(1)

# ---- timing ----
BATCH, C = 512, 1
per_set = 2 * BATCH * C * H * W * 4
nbuf = max(2, (2 * 256 * 1024 * 1024 + per_set - 1) // per_set)
imgs = [torch.rand((BATCH, C, H, W), generator=g, device=dev) * 2 - 1 for _ in range(nbuf)]
ctrls = [ident[None] + 0.05 * (torch.rand((BATCH, F, 2), generator=g, device=dev) * 2 - 1) for _ in range(nbuf)]
outs = [torch.empty((BATCH, C, H, W), device=dev) for _ in range(nbuf)]
SMAX = 4
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(SMAX - 1)]
plans = {}
for npw in (1, 2, 4):
    if HAVE_FLAGS:
        plans[npw] = []
        for j in range(nbuf):
            row = []
            for st in streams:
                with torch.cuda.stream(st):
                    row.append(ops.WarpPlan(imgs[j], ctrls[j], gg.inv_delta_C, gg.P_hat, (H, W), outs[j], P_hat_t=p_hat_t,
                                            table_flags=flags | np_flags(npw)))
            plans[npw].append(row)
if not HAVE_FLAGS:
    base = []
    for j in range(nbuf):
        row = []
        for st in streams:
            with torch.cuda.stream(st):
                row.append(ops.WarpPlan(imgs[j], ctrls[j], gg.inv_delta_C, gg.P_hat, (H, W), outs[j], P_hat_t=p_hat_t, table_flags=flags))
        base.append(row)
    plans = {1: base, 2: base, 4: base}

t_end = time.perf_counter() + 0.4
j = 0
while time.perf_counter() < t_end:
    for _ in range(64):
        outs[j % nbuf].copy_(imgs[j % nbuf]); j += 1
    torch.cuda.synchronize()


def timed(npw, S, first):
    set_np(npw)
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    for k in range(S):
        e0[k].record(streams[k])
    for i in range(K):
        plans[npw][(first + i) % nbuf][i % S].run()
    for k in range(S):
        e1[k].record(streams[k])
    torch.cuda.synchronize()
    return max(b.elapsed_time(e) for e in e1 for b in e0) * 1e3 / K


variants = [(1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3), (2, 4), (4, 2), (4, 3), (4, 4)]
for v in variants:
    timed(v[0], v[1], 0)
res = {v: [] for v in variants}
first = 0
for r in range(R):
    for v in variants:
        torch.cuda.synchronize()
        res[v].append(timed(v[0], v[1], first)); first += K
alg = 76960 * BATCH
print(f"K = {K} launches per region, {R} regions per variant, interleaved; us per 512-image launch (median / min), fraction of 8 TB/s at the median")
for v in variants:
    a = sorted(res[v])
    med = a[(len(a) - 1) // 2]
    print(f"  pairs per workgroup {v[0]}, streams {v[1]}: {med:6.2f} / {a[0]:6.2f} us   {alg / (med * 1e-6) / 8e12:.3f}")
set_np(1)
