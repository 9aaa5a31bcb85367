"""CPU oracle for the TPS++ control-point regressor, the classic localisation network and the
backbone stem (functional restatement over a plain state_dict).

TEST INFRASTRUCTURE, NOT PRODUCT: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  ``tps_pp_amd`` never imports it.

Parity pin: outputs of the reference itself on the inputs of tests/golden/cases.py, committed as
tests/golden/{tpspp_module_v2,tpspp_module_v1,classic_module,backbone_stem}.npz by
tests/golden/make_golden.py (build container) and replayed by tests/test_oracle_golden.py.

The arithmetic here is floating-point convolution / linear algebra; it is restated with PyTorch's
CPU functional ops (``F.conv2d``, ``F.linear``, ``F.layer_norm`` ...) exactly in the order the
reference composes them, not by instantiating modules.  Every function cites the reference lines
it follows (paths relative to /root/reference/mmocr/models/textrecog/).

``sd`` is always a ``dict[str, torch.Tensor]`` with the reference's state_dict keys.
"""
import torch
import torch.nn.functional as F

from . import tps_oracle


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(a)


def rb(x):
    """Round to bfloat16 (nearest even), keep float32 storage."""
    return x.to(torch.bfloat16).float()


def conv_module(sd, prefix, x, stride=1, padding=0, bf16=False):
    """mmcv ConvModule as used by the reference: conv (+bias) then ReLU
    (backbones/tps_pp/tps_pp.py:126-131,149-154,538-552).
    bf16=True restates the bf16 configuration of the build (BASELINE.json configs[2]): operands rounded to
    bfloat16, products and sums in fp32, bias / ReLU in fp32; the CALLER rounds the result where the build
    stores it as bfloat16."""
    w = sd[prefix + ".conv.weight"]
    if bf16:
        x, w = rb(x), rb(w)
    return F.relu(F.conv2d(x, w, sd[prefix + ".conv.bias"], stride=stride, padding=padding))


def cbam(sd, prefix, x):
    """CBAM (tps_pp.py:27-82): channel attention (shared 1x1 MLP on avg- and max-pooled maps),
    then spatial attention (3x3 conv on [mean_c, max_c])."""
    w0, w2 = sd[prefix + ".channel_attention.shared_MLP.0.weight"], \
        sd[prefix + ".channel_attention.shared_MLP.2.weight"]

    def mlp(v):
        return F.conv2d(F.relu(F.conv2d(v, w0)), w2)
    avg = F.adaptive_avg_pool2d(x, 1)
    mx = F.adaptive_max_pool2d(x, 1)
    out = torch.sigmoid(mlp(avg) + mlp(mx)) * x
    a = torch.mean(out, dim=1, keepdim=True)
    m, _ = torch.max(out, dim=1, keepdim=True)
    sa = torch.sigmoid(F.conv2d(torch.cat([a, m], dim=1), sd[prefix + ".spatial_attention.conv2d.weight"],
                                sd[prefix + ".spatial_attention.conv2d.bias"], padding=1))
    return sa * out


def msfa(sd, feat_cat, p_stride=2, bf16=False):
    """Encoder_Decoder_Feature_Extractor.forward (tps_pp.py:156-169) with the layer list of
    :94-119: encoder strides 1, 2, p_stride, (2,1); decoder upsample (2,1), p_stride, 2, 1.
    bf16: every map is stored as bfloat16 except the bottleneck (feeds CBAM and the control points) and
    the last decoder output (feeds DGAB / the score), which stay fp32."""
    inter = {}
    k = feat_cat
    feats = []
    q = rb if bf16 else (lambda v: v)
    for i, st in enumerate([1, 2, p_stride, (2, 1)]):
        k = conv_module(sd, f"MSFA.conv.k_encoder.{i}", k, stride=st, padding=1, bf16=bf16)
        if i < 3:
            k = q(k)
        feats.append(k)
        inter[f"enc{i}"] = k
    point = feats[-1]
    k = cbam(sd, "MSFA.conv.atten", point)
    inter["cbam"] = k
    scales = [(2, 1), p_stride, 2, 1]
    for i in range(3):
        k = F.interpolate(k, scale_factor=scales[i], mode="nearest")
        k = conv_module(sd, f"MSFA.conv.k_decoder.{i}.1", k, padding=1, bf16=bf16)
        k = q(k + feats[2 - i])
        inter[f"dec{i}"] = k
    k = F.interpolate(k, scale_factor=scales[3], mode="nearest")
    k = conv_module(sd, "MSFA.conv.k_decoder.3.1", k, padding=1, bf16=bf16)
    inter["dec3"] = k
    return point, k, inter


def dgab(sd, prefix, x, y, bf16=False):
    """DGAB.forward (backbones/tps_pp/DGAB.py:74-77) with DGAB_Block.forward (:39-55) and Mlp
    (:17-23).  LayerNorm is over the trailing (H, W); proj / fc1 / fc2 act along W.
    bf16: the operands of proj / fc1 / fc2 (gated map, normalised x1, GELU output; weights) rounded to bfloat16,
    everything else fp32 -- the build's bf16 configuration."""
    q = rb if bf16 else (lambda v: v)
    hw = tuple(x.shape[-2:])
    xn = F.layer_norm(x, hw, sd[prefix + ".norm1.weight"], sd[prefix + ".norm1.bias"])
    yt = y.transpose(1, 2)
    w = F.linear(torch.cat([xn.mean(2), yt], 2), sd[prefix + ".attn.mlp_w.0.weight"])
    v_w = w[:, :, :-1].softmax(dim=-1).unsqueeze(2)
    h = F.linear(torch.cat([xn.mean(3), yt], 2), sd[prefix + ".attn.mlp_h.0.weight"])
    v_h = h[:, :, :-1].softmax(dim=-1).unsqueeze(3)
    a = v_h * xn * h[:, :, -1].unsqueeze(-1).unsqueeze(-1) + \
        v_w * xn * w[:, :, -1].unsqueeze(-1).unsqueeze(-1)
    a = F.linear(q(a), q(sd[prefix + ".attn.proj.weight"]), sd[prefix + ".attn.proj.bias"])
    x = x + a
    xn2 = F.layer_norm(x, hw, sd[prefix + ".norm2.weight"], sd[prefix + ".norm2.bias"])
    m = F.linear(q(xn2), q(sd[prefix + ".mlp.fc1.weight"]), sd[prefix + ".mlp.fc1.bias"])
    m = F.gelu(m)
    m = F.linear(q(m), q(sd[prefix + ".mlp.fc2.weight"]), sd[prefix + ".mlp.fc2.bias"])
    return x + m


def tpe(sd, en_feat, de_feat, scale=64 ** -0.5, bf16=False):
    """Transformation_Parameter_Estimation.forward (tps_pp.py:315-325), get_score / atten_score
    (:293-312): control points and tanh attention score."""
    n = en_feat.size(0)
    en = en_feat.flatten(2).transpose(1, 2)
    de = dgab(sd, "TPE.atten.0", de_feat, en, bf16=bf16)
    f1 = F.relu(F.linear(en, sd["TPE.localization_fc1.0.weight"], sd["TPE.localization_fc1.0.bias"]))
    f1 = F.relu(F.linear(f1, sd["TPE.localization_fc1.2.weight"], sd["TPE.localization_fc1.2.bias"]))
    ctrl = F.linear(f1.reshape(n, -1), sd["TPE.localization_fc2.weight"],
                    sd["TPE.localization_fc2.bias"]).view(n, -1, 2)

    def two(prefix, v):
        v = F.linear(v, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"])
        return F.linear(v, sd[prefix + ".1.weight"], sd[prefix + ".1.bias"])
    f = two("TPE.feat_linear", de.flatten(2).transpose(1, 2))
    p = two("TPE.p_linear", en)
    score = torch.tanh(torch.einsum("bmc,bnc->bmn", f, p).mul(scale))
    return ctrl, score, de


def tpspp_regress(sd, x, outs, variant="ResNet45v2", p_stride=2, bf16=False):
    """TPS_PP.forward up to the control points (tps_pp.py:572-594).  Returns
    (ctrl, score, feat_grid, intermediates).
    bf16=True: the build's bf16 configuration -- convolutions on bfloat16 operands with fp32 accumulation,
    feature maps between convolutions (and `feat_grid`, which the warp samples) stored as bfloat16, `en_feat` /
    `de_feat` and everything after them (CBAM, DGAB, control points, score, TPS solve, grid, interpolation) in
    fp32; the caller rounds the two warped outputs to bfloat16 (the module boundary is bf16)."""
    x, outs = _t(x), [_t(o) for o in outs]
    inter = {}
    q = rb if bf16 else (lambda v: v)
    cm = lambda *a, **k: conv_module(*a, bf16=bf16, **k)          # noqa: E731
    if variant == "ResNet45v2":                                   # :580-585
        feat0 = q(cm(sd, "down0", outs[0]))
        feat1 = q(cm(sd, "down1", outs[1]))
        feat2 = q(cm(sd, "down2", x))
        feat_cat = torch.cat((q(cm(sd, "down0_1", feat0, stride=2, padding=1)),
                              q(cm(sd, "down1_1", feat1, stride=2, padding=1)), feat2), dim=1)
        up = F.interpolate(feat2, scale_factor=2, mode="nearest")
        feat_grid = q(cm(sd, "down_feat", torch.cat((feat0, feat1, up), dim=1)))   # :560-562
    else:                                                         # 'ResNet45', :574-579
        feat0 = q(cm(sd, "down0", outs[0], stride=2, padding=1))
        feat1 = q(cm(sd, "down1", outs[1]))
        feat2 = q(cm(sd, "down2", x))
        feat_cat = torch.cat((feat0, feat1, feat2), dim=1)
        feat_grid = x
    inter["feat_cat"], inter["feat_grid"] = feat_cat, feat_grid
    en, de, m_inter = msfa(sd, feat_cat, p_stride, bf16=bf16)
    inter.update(m_inter)
    ctrl, score, de2 = tpe(sd, en, de, bf16=bf16)
    inter["dgab"] = de2
    return ctrl, score, feat_grid, inter


def tpspp_forward(sd, x, outs, variant="ResNet45v2", rectified_img_size=(16, 64), point_size=(2, 16), bf16=False):
    """Whole TPS_PP.forward (tps_pp.py:564-625): regressor (above) + grid + two grid_samples (the C
    oracle).  Returns dict(output, mp_img, pc_score, ctrl, grid)."""
    with torch.no_grad():
        ctrl, score, feat_grid, _ = tpspp_regress(sd, x, outs, variant, bf16=bf16)
    P_xy = sd.get("_P_xy")
    if P_xy is None:
        P_xy = tps_oracle.tpspp_constants(rectified_img_size, point_size)["P_xy"]
    r = tps_oracle.warp(feat_grid.numpy(), ctrl.numpy(), sd["atten_tps.hat_C"].numpy(),
                        sd["atten_tps.P_hat"].numpy(), rectified_img_size, P_xy=P_xy,
                        score=score.numpy(), in1=_t(x).numpy(), want_grid=True)
    return dict(output=r["out0"], mp_img=r["out1"], pc_score=score.numpy(), ctrl=ctrl.numpy(),
                grid=r["grid"])


# ------------------------------------------------------------------------------------------------
def classic_localization(sd, img, prefix="LocalizationNetwork"):
    """LocalizationNetwork.forward (preprocessor/tps_preprocessor.py:141-156), eval-mode BN:
    4 x [conv3x3 no bias, BN, ReLU, pool] + fc1 (ReLU) + fc2 -> (N, F, 2)."""
    x = _t(img)
    with torch.no_grad():
        for ci, bi, pool in ((0, 1, "max"), (4, 5, "max"), (8, 9, "max"), (12, 13, "avg")):
            x = F.conv2d(x, sd[f"{prefix}.conv.{ci}.weight"], None, stride=1, padding=1)
            x = F.batch_norm(x, sd[f"{prefix}.conv.{bi}.running_mean"], sd[f"{prefix}.conv.{bi}.running_var"],
                             sd[f"{prefix}.conv.{bi}.weight"], sd[f"{prefix}.conv.{bi}.bias"], False, 0.1, 1e-5)
            x = F.relu(x)
            x = F.max_pool2d(x, 2, 2) if pool == "max" else F.adaptive_avg_pool2d(x, 1)
        n = x.size(0)
        x = x.view(n, -1)
        x = F.relu(F.linear(x, sd[f"{prefix}.localization_fc1.0.weight"], sd[f"{prefix}.localization_fc1.0.bias"]))
        x = F.linear(x, sd[f"{prefix}.localization_fc2.weight"], sd[f"{prefix}.localization_fc2.bias"])
    return x.view(n, -1, 2)


def classic_forward(sd, img, rectified_img_size=(32, 100)):
    """TPSPreprocessor.forward (tps_preprocessor.py:60-85)."""
    ctrl = classic_localization(sd, img).numpy()
    r = tps_oracle.warp(_t(img).numpy(), ctrl, sd["GridGenerator.inv_delta_C"].numpy(),
                        sd["GridGenerator.P_hat"].numpy(), rectified_img_size, want_grid=True)
    return dict(ctrl=ctrl, grid=r["grid"], out=r["out0"])


# ------------------------------------------------------------------------------------------------
def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.1, 1e-5)


def basic_block(sd, prefix, x, stride):
    """layers/conv_layer.py:12-33 on mmcv's BasicBlock.forward: conv1 (1x1) - bn1 - relu -
    conv2 (3x3, stride) - bn2 - (+ downsample(x)) - relu."""
    out = F.relu(_bn(sd, prefix + ".bn1", F.conv2d(x, sd[prefix + ".conv1.weight"])))
    out = _bn(sd, prefix + ".bn2", F.conv2d(out, sd[prefix + ".conv2.weight"], stride=stride, padding=1))
    res = x
    if prefix + ".downsample.0.weight" in sd:
        res = _bn(sd, prefix + ".downsample.1",
                  F.conv2d(x, sd[prefix + ".downsample.0.weight"], stride=stride))
    return F.relu(out + res)


def backbone_stem(sd, img, arch=(3, 4), strides=(1, 2)):
    """ResNetABI_v2_large.forward up to the TPS call site (backbones/resnet_v2_large.py:173-191):
    conv1 (3x3, bias) - bn1 - relu, then layer1, layer2.  Returns (x, outs) exactly as handed to
    `tpsnet(x, outs)`."""
    x = _t(img)
    with torch.no_grad():
        x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1)))
        outs = []
        for li, (nb, st) in enumerate(zip(arch, strides)):
            outs.append(x)
            for b in range(nb):
                x = basic_block(sd, f"layer{li + 1}.{b}", x, st if b == 0 else 1)
    return x, outs


def backbone_forward(sd, img, arch=(3, 4, 6, 6, 3), strides=(2, 1, 2, 1, 2), tps_sd=None, variant="ResNet45"):
    """Whole ResNetABI_v2_large.forward (backbones/resnet_v2_large.py:162-196): stem, then the five
    stages with the TPS++ network applied to the input of stage index 2 (`outputs = tpsnet(x, outs)`,
    `x = outputs['output']`).  Returns dict(output, img_ref) like the reference."""
    x = _t(img)
    img_ref = None
    with torch.no_grad():
        x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1)))
        outs = []
        for li, (nb, st) in enumerate(zip(arch, strides)):
            if li == 2 and tps_sd is not None:
                r = tpspp_forward(tps_sd, x, outs, variant)
                x = torch.from_numpy(r["output"])
                img_ref = x
            outs.append(x)
            for b in range(nb):
                x = basic_block(sd, f"layer{li + 1}.{b}", x, st if b == 0 else 1)
    return dict(output=x, img_ref=img_ref)


def recognizer_simple_test(backbone_sd, tps_sd, enc_sd, dec_sd, img, resize_widths=None, variant="ResNet45",
                           n_head=8, max_seq_len=40):
    """EncodeDecodeRecognizer.simple_test (recognizer/encode_decode_recognizer.py:186-221) for the
    NRTR + TPS++ config (configs/textrecog/nrtr/nrtr_tps++.py:26-42): valid_ratio = resize width / batch
    width, backbone (+TPS++) -> encoder -> greedy decoder -> AttnConvertor."""
    from . import nrtr_oracle as NO
    img = _t(img)
    ratios = None if resize_widths is None else [1.0 * w / img.shape[-1] for w in resize_widths]
    feat = backbone_forward(backbone_sd, img, tps_sd=tps_sd, variant=variant)["output"]
    r = NO.head_simple_test(enc_sd, dec_sd, feat, n_head, max_seq_len, ratios)
    r["feat"] = feat
    return r
