"""extract_feat (backbone + TPS++) in the bf16 configuration, batch 512: for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P  # noqa: E402

dev = torch.device("cuda:0")
m = P.build_detector(dict(
    type="NRTR", backbone=dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3], strides=[2, 1, 2, 1, 2]),
    tpsnet=dict(type="TPS_PP", variant="ResNet45"), encoder=dict(type="NRTREncoder"),
    decoder=dict(type="NRTRDecoder"), loss=dict(type="TFLoss"),
    label_convertor=dict(type="AttnConvertor", dict_type="DICT90", with_unknown=True), max_seq_len=40)).eval().to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m.backbone.compute_dtype = torch.bfloat16 if mode == "bf16" else mode
if mode == "bf16x3":
    m.tpsnet.compute_dtype = "bf16x3"
img = torch.rand(512, 3, 32, 128, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(5):
        m.extract_feat(img, test=True)
torch.cuda.synchronize()
