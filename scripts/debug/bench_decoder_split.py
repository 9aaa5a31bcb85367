"""Greedy decoder (batch 512) as ONE call against G independent calls of batch 512 / G on G streams, each enqueued from
its own host thread (ctypes releases the GIL): do the latency-bound step launches of independent image groups overlap?
`python scripts/debug/bench_decoder_split.py [N] [groups ...]`"""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P  # noqa: E402,F401
from tps_pp_amd.nrtr_head import NRTRDecoder  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
groups = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
torch.manual_seed(0)
dec = NRTRDecoder(num_classes=93, max_seq_len=40, start_idx=91, padding_idx=92).eval().to(dev)
out_enc = torch.randn(N, 64, 512, device=dev)
feat = torch.empty(N, 512, 8, 8, device=dev)


def run(G, iters=5):
    streams = [torch.cuda.Stream() for _ in range(G)]
    parts = out_enc.chunk(G)
    res = [None] * G

    def work(g):
        with torch.no_grad(), torch.cuda.stream(streams[g]):
            res[g] = dec(feat[: parts[g].shape[0]], parts[g], None, None, train_mode=False)

    def once():
        if G == 1:
            work(0)
            return
        th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    for _ in range(2):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        once()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e3
    return e0.elapsed_time(e1) / iters, wall, torch.cat([r for r in res])


for tag, cd in (("fp32", None), ("bf16x3", "bf16x3"), ("bf16", torch.bfloat16)):
    dec.compute_dtype = cd
    ref = None
    for G in groups:
        ms, wall, out = run(G)
        if ref is None:
            ref = out
        same = (out.argmax(-1) == ref.argmax(-1)).float().mean().item()
        print(f"decoder {tag:7s} batch {N} as {G} group(s): {ms:6.2f} ms (host wall {wall:6.2f} ms)  tokens equal to 1 group: {same:.4f}")
