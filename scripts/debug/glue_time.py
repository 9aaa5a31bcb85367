"""Where the recogniser's batch time goes OUTSIDE its three stages (round-5 review, "what's weak" 8): host time stamps around
every call of `simple_test` and the device time of the same iteration.  python scripts/debug/glue_time.py [bf16|bf16x3|fp32]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P
import bench
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(11)
m = P.build_detector(bench.NRTR_TPSPP_MODEL).eval().to(dev)
m.set_compute_dtype({"bf16": torch.bfloat16, "bf16x3": "bf16x3", "fp32": None}[mode])
n = 512
img = torch.rand((n, 3, 32, 128), device=dev) * 2 - 1
metas = [dict(resize_shape=(32, 128, 3)) for _ in range(n)]
sync = torch.cuda.synchronize
with torch.no_grad():
    for _ in range(3):
        m(img, metas, return_loss=False)
    sync()
    rows = []
    for it in range(6):
        t = [time.perf_counter()]
        for mm in metas:
            mm["valid_ratio"] = 1.0 * mm["resize_shape"][1] / img.size(-1)
        t.append(time.perf_counter())
        feat = m.extract_feat(img, test=True)["output"]; t.append(time.perf_counter())
        out_enc = m.encoder(feat, metas); t.append(time.perf_counter())
        out_dec = m.decoder(feat, out_enc, None, metas, train_mode=False); t.append(time.perf_counter())
        idx, sc = m.label_convertor.tensor2idx(out_dec, metas); t.append(time.perf_counter())
        strs = m.label_convertor.idx2str(idx); t.append(time.perf_counter())
        res = [dict(text=s, score=c) for s, c in zip(strs, sc)]; t.append(time.perf_counter())
        rows.append([1e3 * (b - a) for a, b in zip(t, t[1:])])
    names = ["metas", "feat(host)", "enc(host)", "dec(host)", "tensor2idx(sync)", "idx2str", "dicts"]
    for r in rows[1:]:
        print("  ".join(f"{k} {v:.3f}" for k, v in zip(names, r)), f" total {sum(r):.3f} ms")
    # device-only time of the three stages enqueued back to back, no host sync in between
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync(); e0.record()
    for _ in range(3):
        feat = m.extract_feat(img, test=True)["output"]; out_enc = m.encoder(feat, metas)
        out_dec = m.decoder(feat, out_enc, None, metas, train_mode=False)
    e1.record(); sync()
    print(f"three stages back to back, no host sync: {e0.elapsed_time(e1) / 3:.3f} ms per batch")
    t0 = time.perf_counter()
    for _ in range(3):
        m(img, metas, return_loss=False)
    sync()
    print(f"simple_test end to end: {1e3 * (time.perf_counter() - t0) / 3:.3f} ms per batch")
