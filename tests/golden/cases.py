"""Input definitions of the golden cases.

Inputs are never stored: they are regenerated, bit for bit, from `tps_pp_amd.synth` (a counter
hash, no RNG streams).  `make_golden.py` (build container, has /root/reference) feeds them to the
reference and stores only the reference's OUTPUTS in the .npz files next to this module; the tests
regenerate the same inputs and compare oracle / HIP results with the stored outputs.
Nothing here touches /root/reference.
"""
import os

import numpy as np

from tps_pp_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return np.load(os.path.join(HERE, name + ".npz"))


def host_fingerprint():
    """What decides which MKL / oneDNN kernels PyTorch-CPU picks: CPU model, torch build, thread count."""
    import torch
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"cpu": model, "torch": torch.__version__, "threads": torch.get_num_threads()}


def on_generating_host():
    """True on a host like the one make_golden.py ran on (HOST.json): there the PyTorch-CPU oracles must
    reproduce the fixtures bit for bit; elsewhere they are held to rounding."""
    import json
    try:
        with open(os.path.join(HERE, "HOST.json")) as f:
            return json.load(f) == host_fingerprint()
    except OSError:
        return False


# ---- G2: classic grid + sampler, F=20, 32x100 -> 32x100, N=8 -----------------------------------
CL_F, CL_HW, CL_C, CL_N = 20, (32, 100), 3, 8
# per-image perturbation of the control points: 0 = the module's initial lattice, large values
# push much of the grid outside [-1,1] and exercise the border clamp.
CL_PERTURB = np.array([0.0, 0.01, 0.05, 0.1, 0.3, 1.0, 2.0, 0.05], dtype=np.float32)


def classic_initial_ctrl(F=CL_F):
    """(F,2) fp32: bias of localization_fc2 -- values only, restated in oracle/tps_oracle.py too."""
    half = F // 2
    x = np.linspace(-1.0, 1.0, half)
    top = np.stack([x, np.linspace(0.0, -1.0, num=half)], axis=1)
    bot = np.stack([x, np.linspace(1.0, 0.0, num=half)], axis=1)
    return np.concatenate([top, bot], axis=0).astype(np.float32)


def classic_identity_ctrl(F=CL_F):
    """(F,2) fp32: the fiducial lattice C itself (top edge y=-1, bottom edge y=+1): C' = C makes
    the TPS the identity map."""
    half = F // 2
    x = np.linspace(-1.0, 1.0, half)
    return np.concatenate([np.stack([x, -np.ones(half)], axis=1),
                           np.stack([x, np.ones(half)], axis=1)], axis=0).astype(np.float32)


def g2_inputs():
    ctrl = classic_initial_ctrl()[None] + \
        CL_PERTURB[:, None, None] * synth.dyadic((CL_N, CL_F, 2), "g2.ctrl", 2)
    # image 7: identity lattice + small noise (the bench workload's control points)
    ctrl[7] = classic_identity_ctrl() + 0.05 * synth.dyadic((CL_F, 2), "g2.ctrl7", 2)
    img = synth.dyadic((CL_N, CL_C) + CL_HW, "g2.img", 2)
    img_smooth = synth.smooth_image((CL_N, CL_C) + CL_HW, "g2.img_smooth", 2)
    return dict(ctrl=ctrl.astype(np.float32), img=img, img_smooth=img_smooth)


# ---- G3: TPS_PP warp stage, F=32 (2x16), 16x64, N=2 --------------------------------------------
PP_POINT, PP_HW, PP_C, PP_N = (2, 16), (16, 64), 64, 2
PP_F = PP_POINT[0] * PP_POINT[1]
PP_PERTURB = np.array([0.02, 0.2], dtype=np.float32)


def tpspp_initial_ctrl(point_size=PP_POINT):
    py, px = point_size
    x = np.linspace(0.1, px - 0.1, num=int(px)) / px
    y = np.linspace(0.1, py - 0.1, num=int(py)) / py
    return np.stack(np.meshgrid(x, y), axis=2).reshape(-1, 2).astype(np.float32)


def g3_inputs():
    n = PP_HW[0] * PP_HW[1]
    ctrl = tpspp_initial_ctrl()[None] + \
        PP_PERTURB[:, None, None] * synth.dyadic((PP_N, PP_F, 2), "g3.ctrl", 3)
    score = synth.dyadic((PP_N, n, PP_F), "g3.score", 3)
    feat_grid = synth.smooth_image((PP_N, PP_C, 2 * PP_HW[0], 2 * PP_HW[1]), "g3.feat_grid", 3)
    x = synth.dyadic((PP_N, PP_C) + PP_HW, "g3.x", 3)
    return dict(ctrl=ctrl.astype(np.float32), score=score, feat_grid=feat_grid, x=x)


# ---- G1: classic module (localization CNN + grid + sampler), config[0]: N=4, 3x32x100 ----------
G1_N = 4


def bn_rule(name, shape):
    """synth.state_dict_like override: BatchNorm statistics / affine need sane positive values."""
    if name.endswith("running_var"):
        return (0.25, 1.0)            # in [0.75, 1.25)
    if name.endswith("running_mean"):
        return (0.1, 0.0)
    if name.endswith("num_batches_tracked"):
        return (0.0, 0.0)
    return None


def g1_inputs():
    return dict(img=synth.smooth_image((G1_N, 3, 32, 100), "g1.img", 1))


def g1_state_rule(name, shape):
    r = bn_rule(name, shape)
    if r is not None:
        return r
    if name.endswith("localization_fc2.weight"):
        return (0.02, 0.0)            # keep C' near the initial lattice (reference inits it to 0)
    if ".conv." in name and len(shape) == 1 and name.endswith(".weight"):
        return (0.25, 1.0)            # BatchNorm weight (gamma)
    return None


# keys left at the module's own initial value (the fiducial lattice lives in this bias)
G1_KEEP = ("LocalizationNetwork.localization_fc2.bias", "GridGenerator.inv_delta_C",
           "GridGenerator.P_hat")


def synth_state(module_state, seed, rule=None, keep=()):
    """name -> np.ndarray for every float tensor of a state_dict except `keep` (and integer
    buffers such as num_batches_tracked, which are left alone)."""
    shapes = {k: tuple(v.shape) for k, v in module_state.items()
              if k not in keep and not k.endswith("num_batches_tracked")}
    return synth.state_dict_like(shapes, seed, rule)


# ---- G4 / G5: full TPS_PP module, v2 wiring (reference default) and v1 wiring, N=2 --------------
G4_N = 2
TPSPP_KEEP = ("TPE.localization_fc2.bias", "atten_tps.hat_C", "atten_tps.P_hat")


def tpspp_state_rule(name, shape):
    if name.endswith("localization_fc2.weight"):
        return (0.02, 0.0)            # keep C' near the initial lattice (reference inits it to 0)
    if ".norm1.weight" in name or ".norm2.weight" in name:
        return (0.25, 1.0)            # LayerNorm gamma ~ 1
    if ".norm1.bias" in name or ".norm2.bias" in name:
        return (0.1, 0.0)
    return None


def g4_inputs(variant="ResNet45v2"):
    """x = stage-2 input (N,64,16,64); outs = [stage-0 input, stage-1 input] at the variant's
    geometry (resnet_v2_large.py:183-191).  Post-ReLU features: non-negative."""
    tag = "g4" if variant == "ResNet45v2" else "g5"
    x = np.abs(synth.smooth_image((G4_N, 64, 16, 64), tag + ".x", 4))
    o0 = np.abs(synth.smooth_image((G4_N, 32, 32, 128), tag + ".o0", 4))
    o1_hw = (32, 128) if variant == "ResNet45v2" else (16, 64)
    o1 = np.abs(synth.smooth_image((G4_N, 32) + o1_hw, tag + ".o1", 4))
    return dict(x=x, outs=[o0, o1])


def sub(a):
    """Channel-subsampled view stored for the big intermediates (every 8th channel)."""
    return np.ascontiguousarray(a[:, ::8])


# ---- G7: backbone stem + layer1 + layer2 (eval-mode BN), N=2, 3x32x128 ---------------------------
G7_N = 2
G7_STRIDES = [1, 2, 2, 1, 2]      # the geometry TPS_PP's default wiring needs


def backbone_state_rule(name, shape):
    r = bn_rule(name, shape)
    if r is not None:
        return r
    if (".bn" in name or name.startswith("bn") or ".downsample.1." in name) and name.endswith(".weight"):
        return (0.25, 1.0)            # BatchNorm gamma
    return None


def g7_inputs():
    return dict(img=synth.smooth_image((G7_N, 3, 32, 128), "g7.img", 7))


# ---- G8: NRTRModalityTransform (upstream NRTR conv stem), N=2, 3x32x100 -----------------------------
G8_N = 2


def nrtr_state_rule(name, shape):
    r = bn_rule(name, shape)
    if r is not None:
        return r
    if name.startswith("bn_") and name.endswith(".weight"):
        return (0.25, 1.0)
    return None


def g8_inputs():
    return dict(img=synth.smooth_image((G8_N, 3, 32, 100), "g8.img", 8))


# ---- G9 / G10: recogniser head, small: 2 layers, 2 heads of 64 ------
HD_SMALL = dict(n_layers=2, n_head=2, d_k=64, d_v=64, d_model=128, d_inner=64)
HD_N, HD_HW = 3, (2, 10)
HD_RATIOS = [1.0, 0.6, 0.33]          # valid_ratio per image: key masks of 20, 12 and 7 tokens
HD_MAXLEN = 8
HD_KEEP = ("position_enc.position_table",)
START_IDX, END_IDX, PAD_IDX, NUM_CLASSES = 91, 91, 92, 93   # AttnConvertor(DICT90, with_unknown=True)


def head_state_rule(name, shape):
    if ("norm" in name) and name.endswith(".weight"):
        return (0.25, 1.0)            # LayerNorm gamma ~ 1
    if name == "trg_word_emb.weight":
        return (0.5, 0.0)
    if name == "classifier.weight":
        return (4.0 / np.sqrt(shape[1]), 0.0)   # spread the logits: greedy arg-max far from ties
    return None


def g9_inputs(d_model=HD_SMALL["d_model"], n=HD_N, hw=HD_HW, tag="g9"):
    return dict(feat=synth.dyadic((n, d_model) + tuple(hw), tag + ".feat", 9))


def g10_inputs(d_model=HD_SMALL["d_model"], n=HD_N, t=HD_HW[0] * HD_HW[1]):
    out_enc = synth.dyadic((n, t, d_model), "g10.out_enc", 10)
    # teacher-forcing targets: <BOS>, a few characters, <EOS>, <PAD>...
    body = ((synth.dyadic((n, HD_MAXLEN), "g10.tok", 10) + 1.0) * 45.0).astype(np.int64) % 90
    tgt = np.full((n, HD_MAXLEN), PAD_IDX, dtype=np.int64)
    for i, ln in enumerate([5, 3, 6][:n]):
        tgt[i, 0] = START_IDX
        tgt[i, 1:1 + ln] = body[i, :ln]
        tgt[i, 1 + ln] = END_IDX
    return dict(out_enc=out_enc, padded_targets=tgt)


# ---- G11: the head at the reference's default size (6+6 layers, 8 heads, d_model 512), N=2 --------
G11_N, G11_HW = 2, (4, 16)            # backbone output of a 32x128 image with strides [2,1,2,1,2]


def g11_inputs():
    return dict(feat=synth.dyadic((G11_N, 512) + G11_HW, "g11.feat", 11))


# ---- G12: the whole recogniser (backbone + TPS++ + encoder + decoder + convertor), N=2, 3x32x128 ---
G12_N = 2
G12_STRIDES = [2, 1, 2, 1, 2]         # configs/textrecog/nrtr/nrtr_tps++.py:36
G12_WIDTHS = [128, 96]                # img_meta['resize_shape'][1] of the two images


def g12_inputs():
    return dict(img=synth.smooth_image((G12_N, 3, 32, 128), "g12.img", 12))


# ---- G14: backward of the warp (row F2): upstream gradients for the G2 / G3 forward cases ----------
def g14_inputs():
    return dict(g_out_cl=synth.dyadic((CL_N, CL_C) + CL_HW, "g14.gout_cl", 14),
                g_out0=synth.dyadic((PP_N, PP_C) + PP_HW, "g14.gout0", 14),
                g_out1=synth.dyadic((PP_N, PP_C) + PP_HW, "g14.gout1", 14))
