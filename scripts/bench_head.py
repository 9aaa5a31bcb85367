"""Whole-recogniser inference on the GPU box (BASELINE.json configs[3]/[4] shape: NRTR + TPS++,
3x32x128 images): stage split and images/s.  Random-init weights, synthetic images."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tps_pp_amd as P  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


torch.manual_seed(0)
m = P.build_detector(dict(
    type="NRTR", backbone=dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3], strides=[2, 1, 2, 1, 2]),
    tpsnet=dict(type="TPS_PP", variant="ResNet45"), encoder=dict(type="NRTREncoder"),
    decoder=dict(type="NRTRDecoder"), loss=dict(type="TFLoss"),
    label_convertor=dict(type="AttnConvertor", dict_type="DICT90", with_unknown=True), max_seq_len=40)).eval().to(dev)
img = torch.rand(N, 3, 32, 128, device=dev) * 2 - 1
metas = [dict(resize_shape=(32, 128, 3)) for _ in range(N)]
with torch.no_grad():
    t_all = timeit(lambda: m(img, metas, return_loss=False))
    t_feat = timeit(lambda: m.extract_feat(img, test=True))
    feat = m.extract_feat(img, test=True)["output"]
    t_enc = timeit(lambda: m.encoder(feat, None))
    out_enc = m.encoder(feat, None)
    t_dec = timeit(lambda: m.decoder(feat, out_enc, None, None, train_mode=False))
print(f"NRTR+TPS++ batch {N}: simple_test {t_all:.1f} ms = {N / t_all * 1e3:,.0f} img/s | backbone+TPS++ {t_feat:.1f} ms | "
      f"encoder {t_enc:.2f} ms | greedy decoder (40 steps) {t_dec:.1f} ms")
for tag, bb, hd in (("bf16x3 backbone/TPS++ (fp32 tensors, <= 1e-4)", "bf16x3", None),
                    ("bf16x3 backbone/TPS++ and head (fp32 tensors, <= 1e-4)", "bf16x3", "bf16x3"),
                    ("bf16 backbone/TPS++", torch.bfloat16, None),
                    ("bf16 backbone/TPS++ and head", torch.bfloat16, torch.bfloat16)):
    m.backbone.compute_dtype = bb
    m.tpsnet.compute_dtype = bb if bb == "bf16x3" else None
    m.encoder.compute_dtype = m.decoder.compute_dtype = hd
    with torch.no_grad():
        t_all = timeit(lambda: m(img, metas, return_loss=False))
        t_feat = timeit(lambda: m.extract_feat(img, test=True))
        feat = m.extract_feat(img, test=True)["output"]
        t_enc = timeit(lambda: m.encoder(feat, None))
        out_enc = m.encoder(feat, None)
        t_dec = timeit(lambda: m.decoder(feat, out_enc, None, None, train_mode=False))
    print(f"  {tag}: simple_test {t_all:.1f} ms = {N / t_all * 1e3:,.0f} img/s | backbone+TPS++ {t_feat:.1f} ms | "
          f"encoder {t_enc:.2f} ms | greedy decoder {t_dec:.1f} ms")
