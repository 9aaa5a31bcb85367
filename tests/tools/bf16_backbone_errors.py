"""Prints the distance of the bf16 backbone (HIP) from the reference's fp32 feature map / strings (golden G12)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
from test_gpu_head import build_recognizer, dev  # noqa: E402

cuda = torch.device("cuda:0")
G = cases.load("recognizer_e2e")
m = build_recognizer(cuda)
img = dev(cases.g12_inputs()["img"], cuda)
metas = [dict(resize_shape=(32, w, 3)) for w in cases.G12_WIDTHS]
for dt in (None, torch.bfloat16):
    m.backbone.compute_dtype = dt
    with torch.no_grad():
        res = m(img, [dict(mm) for mm in metas], return_loss=False)
        feat = m.extract_feat(img, test=True)["output"]
    ref = G["feat_sub"]
    err = np.abs(feat.float().cpu().numpy()[:, ::8] - ref)
    print(dt, "feat max err", err.max(), "mean", err.mean(), "scale", np.abs(ref).max())
    print("   strings equal:", [r["text"] for r in res] == [str(s) for s in G["text"]])
    print("   ", [r["text"] for r in res][:3], [str(s) for s in G["text"]][:3])
