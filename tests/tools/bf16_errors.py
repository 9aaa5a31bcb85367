"""Prints how far the bf16 configuration of TPS_PP (HIP) is from (a) the bf16-emulating CPU oracle and
(b) the fp32 oracle, on the golden inputs G4 / G5.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
from oracle import tpspp_oracle as TO  # noqa: E402
from tps_pp_amd import TPS_PP  # noqa: E402

cuda = torch.device("cuda:0")
for variant in ("ResNet45v2", "ResNet45"):
    m = TPS_PP(variant=variant).eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    cpu_sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(cuda)
    inp = cases.g4_inputs(variant)
    rb = lambda a: torch.from_numpy(a).to(torch.bfloat16)      # noqa: E731
    x, outs = rb(inp["x"]), [rb(o) for o in inp["outs"]]
    with torch.no_grad():
        ctrl, score, feat_grid = m.regress(x.to(cuda), [o.to(cuda) for o in outs])
        r = m(x.to(cuda), [o.to(cuda) for o in outs])
    ob = TO.tpspp_forward(cpu_sd, x.float().numpy(), [o.float().numpy() for o in outs], variant, bf16=True)
    of = TO.tpspp_forward(cpu_sd, x.float().numpy(), [o.float().numpy() for o in outs], variant, bf16=False)
    for tag, o in (("bf16-oracle", ob), ("fp32-oracle", of)):
        e_ctrl = np.abs(ctrl.cpu().numpy() - o["ctrl"]).max()
        e_score = np.abs(score.float().cpu().numpy() - o["pc_score"]).max()
        for k in ("output", "mp_img"):
            g = r[k].float().cpu().numpy()
            d = np.abs(g - o[k])
            print(f"{variant:10s} vs {tag}: {k:7s} max {d.max():.4f} mean {d.mean():.5f} scale {np.abs(o[k]).max():.3f} "
                  f"p99 {np.quantile(d, 0.99):.4f}")
        print(f"{variant:10s} vs {tag}: ctrl max {e_ctrl:.2e}  score max {e_score:.2e}")
