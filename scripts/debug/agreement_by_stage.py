"""Which stage's bf16 arithmetic flips the recogniser's decisions (bench.py's model, 256 images): one stage in bf16 at a time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tps_pp_amd as P  # noqa: E402
from tps_pp_amd import metrics  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(11)
m = P.build_detector(bench.NRTR_TPSPP_MODEL).eval()
with torch.no_grad():
    m.decoder.classifier.weight.mul_(8.0)
m = m.to(dev)
g = torch.Generator(device=dev).manual_seed(12)
img = torch.rand((256, 3, 32, 128), generator=g, device=dev) * 2 - 1
metas = [dict(resize_shape=(32, 128, 3)) for _ in range(256)]
bf = torch.bfloat16
for name, md in (("bf16 everywhere", bf), ("bf16x3 everywhere", "bf16x3"),
                 ("bf16 backbone, fp32 head", dict(backbone=bf)),
                 ("fp32 backbone, bf16 head", dict(encoder=bf, decoder=bf)),
                 ("fp32 backbone, bf16 encoder only", dict(encoder=bf)),
                 ("fp32 backbone, bf16 decoder K/V only", dict(decoder=bf)),
                 ("bf16 backbone, bf16x3 head", dict(backbone=bf, encoder="bf16x3", decoder="bf16x3"))):
    r = metrics.precision_agreement(m, img, metas, md)
    print(f"{name:42s} teacher-forced {r['teacher_forced_argmax_agreement']:.4f}  word {r['greedy_word_agreement']:.4f}  "
          f"char {r['greedy_char_agreement']:.4f}")
