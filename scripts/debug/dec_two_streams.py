"""Greedy decoder on the whole batch in one call against two half-batches enqueued concurrently from two host threads
on two HIP streams (each half has its own workspace: a deep copy of the decoder module holds it)."""
import copy, os, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P
dev = torch.device("cuda:0"); N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
dec = P.NRTRDecoder().eval().to(dev)
dec2 = copy.deepcopy(dec)
T, C = 160, 512
for mode in (None, "bf16x3", torch.bfloat16):
    dec.compute_dtype = dec2.compute_dtype = mode
    enc = torch.randn(N, T, C, device=dev) * 0.5
    halves = [enc[: N // 2].contiguous(), enc[N // 2:].contiguous()]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    def whole():
        return dec(None, enc, None, None, train_mode=False)
    def half(i, out):
        with torch.cuda.stream(streams[i]):
            out[i] = (dec if i == 0 else dec2)(None, halves[i], None, None, train_mode=False)
    def both():
        out = [None, None]
        th = [threading.Thread(target=half, args=(i, out)) for i in range(2)]
        for t_ in th: t_.start()
        for t_ in th: t_.join()
        return out
    with torch.no_grad():
        for fn, name in ((whole, "one call"), (both, "two halves, two streams")):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
            print(f"{str(mode):16s} {name:26s}: {dt:6.2f} ms")
        a = whole(); b = both(); torch.cuda.synchronize()
        print("   same result:", torch.equal(a, torch.cat(b, 0)))
