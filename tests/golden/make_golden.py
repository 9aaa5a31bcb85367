"""Regenerate tests/golden/*.npz from the reference (BUILD CONTAINER ONLY: needs /root/reference).

    python tests/golden/make_golden.py [case ...]

Runs the reference's own modules (loaded by path under the stubs of `_ref_loader.py`) on the
inputs defined in `cases.py` and stores the reference's outputs.  Also re-verifies, on every run,
the two facts the oracle's arithmetic rests on:
  * torch.bmm on the CPU == zero-initialised k-ascending fp32 FMA chain (oracle/tps_oracle.c),
  * F.grid_sample on the CPU == oracle weight_form 2,
both bit for bit, and refuses to write fixtures if either fails.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases  # noqa: E402
from _ref_loader import load_reference  # noqa: E402
from oracle import tps_oracle as O  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def biteq(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  "
          f"[{', '.join(f'{k}{tuple(v.shape)}' for k, v in arrs.items())}]")


def quiet(fn, *a, **k):
    """The reference prints parameter counts from its constructors (tps_pp.py:211,287,557)."""
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def case_constants(R):
    gg = R["tps_preprocessor"].GridGenerator(cases.CL_F, cases.CL_HW)
    at = quiet(R["tps_pp"].Attention_Enhanced_TPS, cases.PP_HW, cases.PP_POINT)
    save("constants",
         classic_C=gg.C, classic_P=gg.P,
         classic_inv_delta_C=gg.inv_delta_C.numpy(), classic_P_hat=gg.P_hat.numpy(),
         pp_C=at.C, pp_P=at.P, pp_hat_C=at.hat_C.numpy(), pp_P_hat=at.P_hat.numpy())


def case_g2(R):
    gg = R["tps_preprocessor"].GridGenerator(cases.CL_F, cases.CL_HW)
    inp = cases.g2_inputs()
    Hr, Wr = cases.CL_HW
    grid = gg.build_P_prime(t(inp["ctrl"]), "cpu")
    g4 = grid.reshape(cases.CL_N, Hr, Wr, 2)
    out = Fn.grid_sample(t(inp["img"]), g4, padding_mode="border", align_corners=True).numpy()
    out_s = Fn.grid_sample(t(inp["img_smooth"]), g4, padding_mode="border",
                           align_corners=True).numpy()
    grid = grid.numpy()
    # pin the oracle's arithmetic
    T = O.solve_T(gg.inv_delta_C.numpy(), inp["ctrl"])
    og = O.build_grid(gg.P_hat.numpy(), T)
    assert biteq(og, grid), "bmm != FMA chain: the oracle's summation order is no longer valid"
    assert biteq(O.grid_sample(inp["img"], og, cases.CL_HW, 2), out), "grid_sample != weight_form 2"
    assert biteq(O.grid_sample(inp["img_smooth"], og, cases.CL_HW, 2), out_s)
    # larger batches go through different MKL paths: check 512 too (not stored)
    from tps_pp_amd import synth
    c512 = cases.classic_initial_ctrl()[None] + 0.05 * synth.dyadic((512, cases.CL_F, 2), "c512")
    assert biteq(O.build_grid(gg.P_hat.numpy(), O.solve_T(gg.inv_delta_C.numpy(), c512)),
                 gg.build_P_prime(t(c512), "cpu").numpy())
    save("classic_warp", grid=grid, out=out, out_smooth=out_s)


def case_g3(R):
    at = quiet(R["tps_pp"].Attention_Enhanced_TPS, cases.PP_HW, cases.PP_POINT)
    inp = cases.g3_inputs()
    Hr, Wr = cases.PP_HW
    grid = at.build_P_prime(t(inp["ctrl"]), t(inp["score"]), "cpu")
    g4 = grid.reshape(cases.PP_N, Hr, Wr, 2)
    output = Fn.grid_sample(t(inp["feat_grid"]), g4, padding_mode="border",
                            align_corners=True).numpy()
    mp_img = Fn.grid_sample(t(inp["x"]), g4, padding_mode="border", align_corners=True).numpy()
    grid = grid.numpy()
    P_xy = at.P.astype(np.float32)
    T = O.solve_T(at.hat_C.numpy(), inp["ctrl"])
    og = O.build_grid(at.P_hat.numpy(), T, P_xy, inp["score"])
    assert biteq(og, grid), "TPS_PP bmm != FMA chain"
    assert biteq(O.grid_sample(inp["feat_grid"], og, cases.PP_HW, 2), output)
    assert biteq(O.grid_sample(inp["x"], og, cases.PP_HW, 2), mp_img)
    save("tpspp_warp", grid=grid, output=output, mp_img=mp_img)


def case_g1(R):
    m = R["tps_preprocessor"].TPSPreprocessor(num_fiducial=cases.CL_F, img_size=cases.CL_HW,
                                              rectified_img_size=cases.CL_HW, num_img_channel=3)
    m.eval()
    sd = cases.synth_state(m.state_dict(), 1, cases.g1_state_rule, cases.G1_KEEP)
    missing = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert set(missing.missing_keys) <= set(cases.G1_KEEP) | {
        k for k in m.state_dict() if k.endswith("num_batches_tracked")}, missing
    img = cases.g1_inputs()["img"]
    ctrl = m.LocalizationNetwork(t(img))
    grid = m.GridGenerator.build_P_prime(ctrl, "cpu")
    out = m(t(img))
    save("classic_module", ctrl=ctrl.numpy(), grid=grid.numpy(), out=out.numpy())


def _tpspp_case(R, variant, fname):
    mod = R["tps_pp"]
    m = quiet(mod.TPS_PP)
    if variant == "ResNet45":
        # the reference hard-codes type='ResNet45v2' (tps_pp.py:522); its other branch is reached
        # by patching the instance exactly as SURVEY.md section 0 fact 4 describes
        m.type = "ResNet45"
        m.down0 = sys.modules["mmcv.cnn"].ConvModule(32, m.img_channel, kernel_size=3, stride=2, padding=1)
        for n in ("down0_1", "down1_1", "down_feat", "up_sample"):
            delattr(m, n)
    m.eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    inp = cases.g4_inputs(variant)
    x, outs = t(inp["x"]), [t(o) for o in inp["outs"]]
    # named intermediates through forward hooks on the reference's own sub-modules
    inter = {}

    def hook(name):
        def f(_m, _i, o):
            inter[name] = (o[0] if isinstance(o, tuple) else o).detach().clone()
        return f
    hs = []
    for i in range(4):
        hs.append(m.MSFA.conv.k_encoder[i].register_forward_hook(hook(f"enc{i}")))
        hs.append(m.MSFA.conv.k_decoder[i].register_forward_hook(hook(f"dec{i}_conv")))
    hs.append(m.MSFA.conv.atten.register_forward_hook(hook("cbam")))
    hs.append(m.TPE.atten[0].register_forward_hook(hook("dgab")))
    hs.append(m.MSFA.register_forward_hook(lambda _m, i, o: inter.__setitem__("feat_cat", i[0].detach().clone())))
    grids = {}
    orig = m.atten_tps.build_P_prime

    def spy(c, s_, d):
        g = orig(c, s_, d)
        grids["ctrl"], grids["grid"] = c.detach().clone(), g.detach().clone()
        return g
    m.atten_tps.build_P_prime = spy
    res = m(x, outs)
    for h in hs:
        h.remove()
    # oracle pin: the functional restatement reproduces the reference bit for bit on this machine
    from oracle import tpspp_oracle as TO
    o = TO.tpspp_forward({k: v for k, v in m.state_dict().items()}, inp["x"], inp["outs"], variant)
    assert biteq(o["ctrl"], grids["ctrl"].numpy()), "regressor oracle != reference (ctrl)"
    assert biteq(o["pc_score"], res["pc_score"].numpy()), "regressor oracle != reference (score)"
    assert biteq(o["grid"], grids["grid"].numpy())
    assert biteq(o["output"], res["output"].numpy()) and biteq(o["mp_img"], res["mp_img"].numpy())
    arrs = dict(ctrl=grids["ctrl"].numpy(), pc_score=res["pc_score"].numpy(), grid=grids["grid"].numpy(),
                output=res["output"].numpy(), mp_img=res["mp_img"].numpy())
    if variant == "ResNet45v2":
        arrs.update(feat_cat_sub=cases.sub(inter["feat_cat"].numpy()), cbam=inter["cbam"].numpy(),
                    dgab_sub=cases.sub(inter["dgab"].numpy()))
        for i in range(4):
            arrs[f"enc{i}_sub"] = cases.sub(inter[f"enc{i}"].numpy())
            arrs[f"dec{i}_conv_sub"] = cases.sub(inter[f"dec{i}_conv"].numpy())
    save(fname, **arrs)
    return m


def case_g4(R):
    m = _tpspp_case(R, "ResNet45v2", "tpspp_module_v2")
    # the state_dict layout itself is part of the drop-in contract
    import json
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump({"TPS_PP": {k: list(v.shape) for k, v in m.state_dict().items()},
                   "TPSPreprocessor(20,(32,100),(32,100),3)": {
                       k: list(v.shape) for k, v in R["tps_preprocessor"].TPSPreprocessor(
                           20, (32, 100), (32, 100), 3).state_dict().items()},
                   "NRTREncoder": {k: list(v.shape)
                                   for k, v in R["nrtr_encoder"].NRTREncoder().state_dict().items()},
                   "NRTRDecoder": {k: list(v.shape) for k, v in R["nrtr_decoder"].NRTRDecoder(
                       num_classes=93, start_idx=91, padding_idx=92, max_seq_len=40).state_dict().items()}},
                  f, indent=1)
    print("  wrote state_dict_keys.json")


def case_g5(R):
    _tpspp_case(R, "ResNet45", "tpspp_module_v1")


def case_g7(R):
    m = R["resnet_v2_large"].ResNetABI_v2_large(strides=cases.G7_STRIDES)
    m.eval()
    full = m.state_dict()
    keep = {k: v for k, v in full.items() if k.startswith(("conv1.", "bn1.", "layer1.", "layer2."))}
    sd = cases.synth_state(keep, 7, cases.backbone_state_rule)
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    img = cases.g7_inputs()["img"]
    got = {}

    class Spy(torch.nn.Module):
        def forward(self, x, outs, **kw):
            got["x"], got["outs"] = x.detach().clone(), [o.detach().clone() for o in outs]
            return {"output": x}
    m(t(img), Spy())
    from oracle import tpspp_oracle as TO
    ox, oouts = TO.backbone_stem({k: v for k, v in m.state_dict().items()}, img)
    assert biteq(ox.numpy(), got["x"].numpy()), "backbone oracle != reference"
    assert all(biteq(a.numpy(), b.numpy()) for a, b in zip(oouts, got["outs"]))
    save("backbone_stem", x=got["x"].numpy(), outs0_sub=np.ascontiguousarray(got["outs"][0].numpy()[:, ::4]),
         outs1_sub=np.ascontiguousarray(got["outs"][1].numpy()[:, ::4]))


def case_g1_pin(R):
    """(no file) the classic-module oracle reproduces the reference on G1."""
    m = R["tps_preprocessor"].TPSPreprocessor(num_fiducial=cases.CL_F, img_size=cases.CL_HW,
                                              rectified_img_size=cases.CL_HW, num_img_channel=3)
    m.eval()
    sd = cases.synth_state(m.state_dict(), 1, cases.g1_state_rule, cases.G1_KEEP)
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    from oracle import tpspp_oracle as TO
    img = cases.g1_inputs()["img"]
    o = TO.classic_forward({k: v for k, v in m.state_dict().items()}, img)
    assert biteq(o["ctrl"], m.LocalizationNetwork(t(img)).numpy())
    assert biteq(o["out"], m(t(img)).numpy())
    print("  classic-module oracle == reference (bitwise)")


def case_g8(R):
    m = R["nrtr_modality_transformer"].NRTRModalityTransform()
    m.eval()
    sd = cases.synth_state(m.state_dict(), 8, cases.nrtr_state_rule)
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    out = m(t(cases.g8_inputs()["img"]))
    save("nrtr_stem", out=out.numpy())


def _head_modules(R, small):
    cfg = dict(cases.HD_SMALL) if small else {}
    enc = R["nrtr_encoder"].NRTREncoder(**cfg).eval()
    dcfg = dict(cfg)
    if small:
        dcfg["d_embedding"] = cfg["d_model"]
    # the recogniser passes these four from the convertor (encode_decode_recognizer.py:62-66)
    dcfg.update(num_classes=cases.NUM_CLASSES, start_idx=cases.START_IDX, padding_idx=cases.PAD_IDX,
                max_seq_len=cases.HD_MAXLEN if small else 40)
    dec = R["nrtr_decoder"].NRTRDecoder(**dcfg).eval()
    for m, seed in ((enc, 9), (dec, 10)):
        sd = cases.synth_state(m.state_dict(), seed, cases.head_state_rule, cases.HD_KEEP)
        m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    return enc, dec


def case_g9(R):
    """NRTR encoder, small, with and without the valid_ratio key mask."""
    from oracle import nrtr_oracle as NO
    enc, _ = _head_modules(R, True)
    feat = cases.g9_inputs()["feat"]
    metas = [dict(valid_ratio=r) for r in cases.HD_RATIOS]
    with torch.no_grad():
        out_m = enc(t(feat), metas)
        out_n = enc(t(feat), None)
    sd = dict(enc.state_dict())
    o_m = NO.encoder_forward(sd, feat, cases.HD_SMALL["n_head"], cases.HD_RATIOS)
    o_n = NO.encoder_forward(sd, feat, cases.HD_SMALL["n_head"], None)
    assert biteq(o_m.numpy(), out_m.numpy()) and biteq(o_n.numpy(), out_n.numpy()), "encoder oracle != reference"
    save("nrtr_encoder", out_masked=out_m.numpy(), out_nomask=out_n.numpy())


def case_g10(R):
    """NRTR decoder, small: teacher-forced logits and greedy decoding scores."""
    from oracle import nrtr_oracle as NO
    _, dec = _head_modules(R, True)
    inp = cases.g10_inputs()
    metas = [dict(valid_ratio=r) for r in cases.HD_RATIOS]
    with torch.no_grad():
        logits = dec(None, t(inp["out_enc"]), dict(padded_targets=t(inp["padded_targets"])), metas, train_mode=True)
        probs = dec(None, t(inp["out_enc"]), None, metas, train_mode=False)
        probs_nomask = dec(None, t(inp["out_enc"]), None, None, train_mode=False)
    sd = dict(dec.state_dict())
    nh = cases.HD_SMALL["n_head"]
    o_l = NO.decoder_forward_train(sd, inp["out_enc"], inp["padded_targets"], nh, cases.PAD_IDX, cases.HD_RATIOS)
    o_p = NO.decoder_forward_test(sd, inp["out_enc"], nh, cases.HD_MAXLEN, cases.START_IDX, cases.PAD_IDX,
                                  cases.HD_RATIOS)
    assert biteq(o_l.numpy(), logits.numpy()) and biteq(o_p.numpy(), probs.numpy()), "decoder oracle != reference"
    assert biteq(NO.sinusoid_table(200, 128).numpy(), sd["position_enc.position_table"].numpy())
    save("nrtr_decoder", logits=logits.numpy(), probs=probs.numpy(), probs_nomask=probs_nomask.numpy())


def case_g11(R):
    """Encoder + greedy decoder + convertor at the reference's default size."""
    from oracle import nrtr_oracle as NO
    enc, dec = _head_modules(R, False)
    conv = R["attn_convertor"].AttnConvertor(dict_type="DICT90", with_unknown=True)
    assert (conv.start_idx, conv.end_idx, conv.padding_idx, conv.num_classes()) == \
        (cases.START_IDX, cases.END_IDX, cases.PAD_IDX, cases.NUM_CLASSES)
    feat = cases.g11_inputs()["feat"]
    with torch.no_grad():
        out_enc = enc(t(feat), None)
        out_dec = dec(None, out_enc, None, None, train_mode=False)
    idx, scores = conv.tensor2idx(out_dec, None)
    text = conv.idx2str(idx)
    o = NO.head_simple_test(dict(enc.state_dict()), dict(dec.state_dict()), feat)
    assert biteq(o["out_enc"].numpy(), out_enc.numpy()) and biteq(o["out_dec"].numpy(), out_dec.numpy())
    assert o["indexes"] == idx and o["text"] == text
    assert NO.attn_dictionary()[0] == conv.idx2char
    tgt = conv.str2tensor(["hello", "W0rld!"])["padded_targets"].numpy()
    print("  texts:", text)
    save("nrtr_head_full", out_enc_sub=np.ascontiguousarray(out_enc.numpy()[:, :, ::8]), out_dec=out_dec.numpy(),
         argmax=out_dec.argmax(-1).numpy().astype(np.int32),
         text=np.array(text), idx_len=np.array([len(i) for i in idx], dtype=np.int32),
         str2tensor_targets=tgt)


def case_g12(R):
    """Whole recogniser: the reference's backbone (calling the reference's TPS_PP, patched to the
    geometry its own NRTR config needs -- SURVEY.md section 0 fact 4), encoder, decoder and convertor,
    composed as EncodeDecodeRecognizer.simple_test composes them (encode_decode_recognizer.py:186-221;
    the recogniser class itself imports modules that were never released)."""
    from oracle import tpspp_oracle as TO
    bb = R["resnet_v2_large"].ResNetABI_v2_large(arch_settings=[3, 4, 6, 6, 3], strides=cases.G12_STRIDES).eval()
    sd = cases.synth_state(bb.state_dict(), 7, cases.backbone_state_rule)
    bb.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    tps = quiet(R["tps_pp"].TPS_PP)
    tps.type = "ResNet45"
    tps.down0 = sys.modules["mmcv.cnn"].ConvModule(32, tps.img_channel, kernel_size=3, stride=2, padding=1)
    for n in ("down0_1", "down1_1", "down_feat", "up_sample"):
        delattr(tps, n)
    tps.eval()
    sd = cases.synth_state(tps.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    tps.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    enc, dec = _head_modules(R, False)
    conv = R["attn_convertor"].AttnConvertor(dict_type="DICT90", with_unknown=True, max_seq_len=40)
    img = t(cases.g12_inputs()["img"])
    metas = [dict(resize_shape=(32, w, 3)) for w in cases.G12_WIDTHS]
    with torch.no_grad():
        for m in metas:                                            # simple_test :196-198
            m["valid_ratio"] = 1.0 * m["resize_shape"][1] / img.size(-1)
        feat = bb(img, tps, True)["output"]
        out_enc = enc(feat, metas)
        out_dec = dec(feat, out_enc, None, metas, train_mode=False)
    idx, scores = conv.tensor2idx(out_dec, metas)
    text = conv.idx2str(idx)
    o = TO.recognizer_simple_test(dict(bb.state_dict()), dict(tps.state_dict()), dict(enc.state_dict()),
                                  dict(dec.state_dict()), img.numpy(), cases.G12_WIDTHS)
    assert biteq(o["feat"].numpy(), feat.numpy()), "backbone oracle != reference"
    assert biteq(o["out_dec"].numpy(), out_dec.numpy()) and o["text"] == text
    p = np.sort(out_dec.numpy(), -1)
    print("  texts:", text, " min arg-max margin:", float((p[..., -1] - p[..., -2]).min()),
          " feat absmax:", float(feat.abs().max()))
    save("recognizer_e2e", feat_sub=np.ascontiguousarray(feat.numpy()[:, ::8]), out_dec=out_dec.numpy(),
         text=np.array(text), score0=np.array(scores[0], dtype=np.float32))


METRIC_PAIRS = [("hello", "hello"), ("Hello", "hello"), ("he-llo!", "hello"), ("helo", "hello"), ("", "abc"),
                ("abc", ""), ("W0rld", "world"), ("kitten", "sitting"), ("ICDAR2015", "icdar-2015"),
                ("~a6666h", "a6666"), ("flaw", "lawn"), ("\u4e2d\u6587ab", "\u4e2d\u6587AB"), ("x", "y")]


def case_g13(R):
    """Word-accuracy / edit-distance metrics from the reference's mmocr/core/evaluation/ocr_metric.py,
    executed in place; rapidfuzz (absent) is stubbed by the textbook unit-cost Levenshtein DP."""
    import json
    import types

    def lev(a, b):
        d = list(range(len(b) + 1))
        for i in range(1, len(a) + 1):
            prev, d[0] = d[0], i
            for j in range(1, len(b) + 1):
                prev, d[j] = d[j], min(d[j] + 1, d[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
        return d[len(b)]
    rf = types.ModuleType("rapidfuzz")
    rf.string_metric = types.SimpleNamespace(levenshtein=lev)
    sys.modules["rapidfuzz"] = rf
    from _ref_loader import _load
    om = _load("mmocr.core.evaluation.ocr_metric", "mmocr/core/evaluation/ocr_metric.py")
    preds, gts = [p for p, _ in METRIC_PAIRS], [g for _, g in METRIC_PAIRS]
    out = dict(pairs=METRIC_PAIRS, count_matches=om.count_matches(preds, gts),
               eval_ocr_metric=om.eval_ocr_metric(preds, gts),
               per_pair=[om.count_matches([p], [g]) for p, g in METRIC_PAIRS])
    with open(os.path.join(HERE, "ocr_metric.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("  wrote ocr_metric.json", out["eval_ocr_metric"], out["count_matches"])


def case_g14(R):
    """Backward of the warp: autograd through the reference's own build_P_prime + F.grid_sample."""
    with torch.enable_grad():
        _case_g14(R)


def _case_g14(R):
    gi = cases.g14_inputs()
    # classic geometry (G2 inputs, smooth image)
    gg = R["tps_preprocessor"].GridGenerator(cases.CL_F, cases.CL_HW)
    inp = cases.g2_inputs()
    img, ctrl = t(inp["img_smooth"]).requires_grad_(True), t(inp["ctrl"]).requires_grad_(True)
    grid = gg.build_P_prime(ctrl, "cpu").reshape(cases.CL_N, cases.CL_HW[0], cases.CL_HW[1], 2)
    (Fn.grid_sample(img, grid, padding_mode="border", align_corners=True) * t(gi["g_out_cl"])).sum().backward()
    o = O.warp_backward(gi["g_out_cl"], inp["img_smooth"], inp["ctrl"], gg.inv_delta_C.numpy(), gg.P_hat.numpy(),
                        cases.CL_HW)
    assert biteq(o["g_in0"], img.grad.numpy()) and biteq(o["g_ctrl"], ctrl.grad.numpy()), "classic bwd oracle"
    arrs = dict(cl_g_img=img.grad.numpy(), cl_g_ctrl=ctrl.grad.numpy())
    # TPS_PP geometry (G3 inputs)
    at = quiet(R["tps_pp"].Attention_Enhanced_TPS, cases.PP_HW, cases.PP_POINT)
    inp = cases.g3_inputs()
    ctrl, score = t(inp["ctrl"]).requires_grad_(True), t(inp["score"]).requires_grad_(True)
    fg, x = t(inp["feat_grid"]).requires_grad_(True), t(inp["x"]).requires_grad_(True)
    grid = at.build_P_prime(ctrl, score, "cpu").reshape(cases.PP_N, cases.PP_HW[0], cases.PP_HW[1], 2)
    loss = (Fn.grid_sample(fg, grid, padding_mode="border", align_corners=True) * t(gi["g_out0"])).sum() + \
        (Fn.grid_sample(x, grid, padding_mode="border", align_corners=True) * t(gi["g_out1"])).sum()
    loss.backward()
    o = O.warp_backward(gi["g_out0"], inp["feat_grid"], inp["ctrl"], at.hat_C.numpy(), at.P_hat.numpy(),
                        cases.PP_HW, P_xy=at.P.astype(np.float32), score=inp["score"], in1=inp["x"],
                        g_out1=gi["g_out1"])
    for k, v in (("g_in0", fg), ("g_in1", x), ("g_ctrl", ctrl), ("g_score", score)):
        assert biteq(o[k], v.grad.numpy()), "TPS_PP bwd oracle: " + k
    arrs.update(pp_g_feat_grid_sub=cases.sub(fg.grad.numpy()), pp_g_x_sub=cases.sub(x.grad.numpy()),
                pp_g_ctrl=ctrl.grad.numpy(), pp_g_score=score.grad.numpy())
    save("warp_backward", **arrs)


CASES = dict(constants=case_constants, g2=case_g2, g3=case_g3, g1=case_g1, g1_pin=case_g1_pin,
             g4=case_g4, g5=case_g5, g7=case_g7, g8=case_g8, g9=case_g9, g10=case_g10, g11=case_g11, g12=case_g12, g13=case_g13, g14=case_g14)


def main(argv):
    R = load_reference()
    for name in (argv or list(CASES)):
        print(f"[{name}]")
        CASES[name](R)
    with open(os.path.join(HERE, "HOST.json"), "w") as f:      # see cases.on_generating_host()
        json.dump(cases.host_fingerprint(), f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
