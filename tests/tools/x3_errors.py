"""Prints the distance of the "bf16x3" configuration (fp32 tensors, three-term bf16 split in the convolutions) from the
reference's fp32 goldens (G4 / G5 module, G12 recogniser) next to the exact-fp32 kernels'.  Run on the GPU box."""
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
from tps_pp_amd import TPS_PP  # noqa: E402

cuda = torch.device("cuda:0")
for variant, fname in (("ResNet45v2", "tpspp_module_v2"), ("ResNet45", "tpspp_module_v1")):
    G = cases.load(fname)
    m = TPS_PP(variant=variant).eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.to(cuda)
    inp = cases.g4_inputs(variant)
    x, outs = dev(inp["x"], cuda), [dev(o, cuda) for o in inp["outs"]]
    for mode in (None, "bf16x3"):
        m.compute_dtype = mode
        with torch.no_grad():
            ctrl, score, _ = m.regress(x, outs)
            r = m(x, outs)
        print(f"{variant:10s} {str(mode):7s}: ctrl {np.abs(ctrl.cpu().numpy() - G['ctrl']).max():.2e}  "
              f"score {np.abs(score.cpu().numpy() - G['pc_score']).max():.2e}  "
              f"output {np.abs(r['output'].cpu().numpy() - G['output']).max():.2e}  "
              f"mp_img {np.abs(r['mp_img'].cpu().numpy() - G['mp_img']).max():.2e}")
G = cases.load("recognizer_e2e")
m = build_recognizer(cuda)
img = dev(cases.g12_inputs()["img"], cuda)
metas = [dict(resize_shape=(32, w, 3)) for w in cases.G12_WIDTHS]
for mode in (None, "bf16x3"):
    m.backbone.compute_dtype = m.tpsnet.compute_dtype = mode
    with torch.no_grad():
        res = m(img, [dict(mm) for mm in metas], return_loss=False)
        feat = m.extract_feat(img, test=True)["output"]
    err = np.abs(feat.cpu().numpy()[:, ::8] - G["feat_sub"])
    print(f"recogniser {str(mode):7s}: feat max err {err.max():.2e} (scale {np.abs(G['feat_sub']).max():.2f})  strings equal "
          f"{[r['text'] for r in res] == [str(s) for s in G['text']]}  score0 err "
          f"{np.abs(np.array(res[0]['score'], dtype=np.float32) - G['score0']).max():.2e}")
