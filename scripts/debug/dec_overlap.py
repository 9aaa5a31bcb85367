"""How much do two half-batch decodes overlap on two streams (two host threads)?"""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = 512
enc = P.NRTREncoder().eval().to(dev)
decs = [P.NRTRDecoder(num_classes=93, start_idx=91, padding_idx=92).eval().to(dev) for _ in range(4)]
for d in decs[1:]:
    d.load_state_dict(decs[0].state_dict())
feat = torch.rand(N, 512, 4, 16, device=dev)
with torch.no_grad():
    out_enc = enc(feat, None)
    full = out_enc.clone()

def run(dec, x, stream, reps):
    with torch.no_grad(), torch.cuda.stream(stream):
        for _ in range(reps):
            dec(None, x, None, None, train_mode=False)

def wall(k, reps=3):
    parts = list(full.chunk(k, 0))
    parts = [p.contiguous() for p in parts]
    streams = [torch.cuda.Stream() for _ in range(k)]
    for i in range(k): run(decs[i], parts[i], streams[i], 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(decs[i], parts[i], streams[i], reps)) for i in range(k)]
    for t in th: t.start()
    for t in th: t.join()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, t_host / reps * 1e3

for k in (1, 2, 4):
    w, h = wall(k)
    print(f"{k} sub-batch(es) of {N // k} on {k} stream(s)/thread(s): {w:.1f} ms per full batch (host enqueue {h:.1f} ms)")
