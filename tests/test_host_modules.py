"""CPU tests of the host-side mirror of the reference's module API (no GPU compute)."""
import json
import os

import numpy as np
import pytest
import torch

import cases
import tps_pp_amd
from tps_pp_amd import (BACKBONES, PREPROCESSOR, TPS_PP, TPSPreprocessor, _lib, build_backbone,
                        build_preprocessor, constants)

KEYS = json.load(open(os.path.join(cases.HERE, "state_dict_keys.json")))


def test_constructor_assertions_like_the_reference():
    """tests/test_models/test_ocr_preprocessor.py:9-17 of the reference."""
    with pytest.raises(AssertionError):
        TPSPreprocessor(num_fiducial=-1)
    with pytest.raises(AssertionError):
        TPSPreprocessor(img_size=32)
    with pytest.raises(AssertionError):
        TPSPreprocessor(rectified_img_size=100)
    with pytest.raises(AssertionError):
        TPSPreprocessor(num_img_channel="bgr")
    with pytest.raises(AssertionError):
        TPS_PP(img_size=[16, 64])
    with pytest.raises(AssertionError):
        TPS_PP(rectified_img_size=64)


def test_registries_build_from_the_reference_configs():
    # configs/textrecog/nrtr/nrtr_tps++.py:38 and configs/_base_/recog_models/crnn_tps.py:7-12
    m = build_backbone(dict(type="TPS_PP"))
    assert isinstance(m, TPS_PP) and m.num_fiducial == 32 and m.rectified_img_size == (16, 64)
    p = build_preprocessor(dict(type="TPSPreprocessor", num_fiducial=20, img_size=(32, 100),
                                rectified_img_size=(32, 100), num_img_channel=1))
    assert isinstance(p, TPSPreprocessor)
    assert "TPS_PPv2" in PREPROCESSOR and "TPS_PP" in BACKBONES
    with pytest.raises(KeyError):
        build_backbone(dict(type="NoSuchThing"))
    with pytest.raises(KeyError):
        build_backbone(dict(num_fiducial=3))


def test_state_dict_layout_matches_the_reference():
    m = TPS_PP()
    sd = m.state_dict()
    assert list(sd) == list(KEYS["TPS_PP"])
    assert all(list(sd[k].shape) == KEYS["TPS_PP"][k] for k in sd)
    p = TPSPreprocessor(20, (32, 100), (32, 100), 3)
    ref = KEYS["TPSPreprocessor(20,(32,100),(32,100),3)"]
    sdp = p.state_dict()
    assert list(sdp) == list(ref) and all(list(sdp[k].shape) == ref[k] for k in sdp)
    # a state_dict in the reference's layout loads strictly
    m.load_state_dict({k: torch.zeros(v) for k, v in KEYS["TPS_PP"].items()}, strict=True)


def test_buffers_and_initial_control_points_match_the_reference():
    K = cases.load("constants")
    m = TPS_PP()
    assert np.array_equal(m.atten_tps.hat_C.numpy(), K["pp_hat_C"])
    assert np.array_equal(m.atten_tps.P_hat.numpy(), K["pp_P_hat"])
    assert np.array_equal(m.atten_tps.P, K["pp_P"]) and np.array_equal(m.atten_tps.C, K["pp_C"])
    assert torch.count_nonzero(m.TPE.localization_fc2.weight) == 0
    assert np.array_equal(m.TPE.localization_fc2.bias.detach().numpy().reshape(-1, 2),
                          cases.tpspp_initial_ctrl())
    p = TPSPreprocessor(20, (32, 100), (32, 100), 1)
    assert np.array_equal(p.GridGenerator.inv_delta_C.numpy(), K["classic_inv_delta_C"])
    assert np.array_equal(p.GridGenerator.P_hat.numpy(), K["classic_P_hat"])
    assert np.array_equal(p.LocalizationNetwork.localization_fc2.bias.detach().numpy().reshape(-1, 2),
                          cases.classic_initial_ctrl())


def test_regressor_matches_oracle_on_cpu():
    """Wiring / state_dict check of the mirror without a GPU: its layers composed with plain PyTorch
    (the explicit test hook, never used by forward()) must agree with the oracle exactly."""
    from oracle import tpspp_oracle as TO
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    inp = cases.g4_inputs()
    with torch.no_grad():
        cp, sc, fg = m._regress_torch(torch.from_numpy(inp["x"]), [torch.from_numpy(o) for o in inp["outs"]])
        ocp, osc, ofg, _ = TO.tpspp_regress(dict(m.state_dict()), inp["x"], inp["outs"])
    assert torch.equal(cp, ocp) and torch.equal(sc, osc) and torch.equal(fg, ofg)
    G = cases.load("tpspp_module_v2")
    np.testing.assert_allclose(cp.numpy(), G["ctrl"], atol=2e-5, rtol=0)


def test_no_cpu_fallback_and_forward_only():
    p = TPSPreprocessor(20, (32, 100), (32, 100), 3)
    with pytest.raises(_lib.TpsppError, match="no CPU fallback"):
        p(torch.zeros(1, 3, 32, 100))                     # grad mode on: the training graph needs the GPU too
    with torch.no_grad(), pytest.raises(_lib.TpsppError, match="no CPU fallback"):
        p(torch.zeros(1, 3, 32, 100))
    m = TPS_PP()
    with torch.no_grad(), pytest.raises(_lib.TpsppError, match="no CPU fallback"):
        m(torch.zeros(1, 64, 16, 64), [torch.zeros(1, 32, 32, 128), torch.zeros(1, 32, 32, 128)])
    from tps_pp_amd import ResNetABI_v2_large
    with torch.no_grad(), pytest.raises(_lib.TpsppError, match="no CPU fallback"):
        ResNetABI_v2_large().eval()(torch.zeros(1, 3, 32, 128))


def test_variant_is_picked_by_geometry_checkpoint_or_backbone_strides():
    """configs/textrecog/nrtr/nrtr_tps++.py:34-38: `tpsnet=dict(type='TPS_PP')` next to backbone strides [2,1,2,1,2] --
    the geometry the reference's hard-coded wiring cannot take."""
    v2_maps = [torch.zeros(1, 32, 32, 128), torch.zeros(1, 32, 32, 128)]
    v1_maps = [torch.zeros(1, 32, 32, 128), torch.zeros(1, 32, 16, 64)]
    x = torch.zeros(1, 64, 16, 64)
    # an explicit choice is never overridden: the other geometry is an error that names the way out
    m = TPS_PP(variant="ResNet45v2")
    with torch.no_grad(), pytest.raises(ValueError, match="variant='ResNet45'"):
        m(x, v1_maps)
    v1 = TPS_PP(variant="ResNet45")
    assert "down0_1.conv.weight" not in v1.state_dict()
    assert tuple(v1.down0.conv.weight.shape) == (64, 32, 3, 3)
    with torch.no_grad(), pytest.raises(ValueError, match="ResNet45v2"):
        v1(x, v2_maps)
    # default: the reference's hard-coded wiring, switched by the first forward with the other geometry ...
    auto = TPS_PP().eval()                                     # (fresh + eval mode only: see the end of this test)
    assert auto.type == "ResNet45v2" and not auto.variant_explicit
    with torch.no_grad(), pytest.raises(_lib.TpsppError, match="no CPU fallback"):      # (then the CPU tensor is refused)
        auto(x, v1_maps)
    assert auto.type == "ResNet45" and list(auto.state_dict()) == list(v1.state_dict())
    # ... by the checkpoint that is loaded (down0.conv.weight is 1x1 in one wiring, 3x3 in the other) ...
    fresh = TPS_PP()
    fresh.load_state_dict(v1.state_dict())
    assert fresh.type == "ResNet45" and torch.equal(fresh.down0.conv.weight, v1.down0.conv.weight)
    with torch.no_grad(), pytest.raises(ValueError, match="set_variant"):      # a loaded module is not re-wired silently
        fresh(x, v2_maps)
    back = TPS_PP()
    back.load_state_dict(TPS_PP(variant="ResNet45v2").state_dict())
    assert back.type == "ResNet45v2"
    with pytest.raises(ValueError, match="checkpoint holds"):
        TPS_PP(variant="ResNet45v2").load_state_dict(v1.state_dict())
    # ... and by the recogniser that owns it, from its backbone's strides
    assert TPS_PP.variant_for_strides([2, 1, 2, 1, 2]) == "ResNet45"
    assert TPS_PP.variant_for_strides([1, 2, 2, 1, 2]) == "ResNet45v2"
    assert TPS_PP.variant_for_strides([2, 2, 1]) is None
    # the 60-entry state_dict order of the reference survives a round trip through the other wiring
    rt = TPS_PP().set_variant("ResNet45", explicit=False).set_variant("ResNet45v2", explicit=False)
    assert list(rt.state_dict()) == list(TPS_PP().state_dict())
    # forward() never re-wires a module that is training (an optimiser may hold its parameters) ...
    tr = TPS_PP().train()
    ids = [id(p) for p in tr.parameters()]
    with pytest.raises(ValueError, match="set_variant"):
        tr(x, v1_maps)
    assert tr.type == "ResNet45v2" and ids == [id(p) for p in tr.parameters()]
    # ... nor one that has loaded anything, a regressor-only (partial) checkpoint included
    part = TPS_PP()
    part.load_state_dict({k: v for k, v in TPS_PP().state_dict().items() if k.startswith("TPE.")}, strict=False)
    with torch.no_grad(), pytest.raises(ValueError, match="set_variant"):
        part(x, v1_maps)
    assert part.type == "ResNet45v2"


def test_eval_mode_keeps_gradients_when_the_inputs_carry_them(monkeypatch):
    """The reference is differentiable in eval mode (saliency / adversarial gradients, frozen-BN fine-tuning of the
    layers upstream): inputs that require grad take the autograd path; plain eval inference does not."""
    m = TPS_PP().eval()
    took = []
    monkeypatch.setattr(m, "_forward_autograd", lambda x, outs: took.append("autograd") or {})
    monkeypatch.setattr(m, "regress", lambda x, outs: (_ for _ in ()).throw(RuntimeError("hip path")))
    x = torch.zeros(1, 64, 16, 64, requires_grad=True)
    maps = [torch.zeros(1, 32, 32, 128), torch.zeros(1, 32, 32, 128)]
    m(x, maps)
    assert took == ["autograd"]
    with torch.no_grad(), pytest.raises(RuntimeError, match="hip path"):
        m(x, maps)                                             # autograd off: inference path
    with pytest.raises(RuntimeError, match="hip path"):
        m(x.detach(), maps)                                    # nothing upstream wants a gradient: inference path (warned once)
    assert m._warned_detached
    from tps_pp_amd import TPSPreprocessor
    pre = TPSPreprocessor(num_fiducial=20, img_size=(32, 100), rectified_img_size=(32, 100), num_img_channel=1).eval()
    with pytest.raises(_lib.TpsppError, match="no CPU fallback"):      # the autograd branch is the one that asks for a GPU tensor first
        pre(torch.zeros(1, 1, 32, 100, requires_grad=True))


def test_register_into_a_mmocr_like_builder(monkeypatch):
    import sys
    import types
    from tps_pp_amd.registry import Registry, register_into_mmocr
    fake = types.ModuleType("mmocr.models.builder")
    fake.BACKBONES, fake.PREPROCESSOR = Registry("models"), Registry("preprocessor")
    pk, pm = types.ModuleType("mmocr"), types.ModuleType("mmocr.models")
    pk.models, pm.builder = pm, fake
    monkeypatch.setitem(sys.modules, "mmocr", pk)
    monkeypatch.setitem(sys.modules, "mmocr.models", pm)
    monkeypatch.setitem(sys.modules, "mmocr.models.builder", fake)
    assert register_into_mmocr() is True
    assert fake.BACKBONES.get("TPS_PP") is TPS_PP
    assert fake.PREPROCESSOR.get("TPSPreprocessor") is TPSPreprocessor


def test_constants_builders_reproduce_reference_tables():
    K = cases.load("constants")
    c = constants.classic(20, (32, 100))
    np.testing.assert_allclose(c["inv_delta_C"], K["classic_inv_delta_C"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c["P_hat"], K["classic_P_hat"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(constants.classic_identity_ctrl(20), cases.classic_identity_ctrl(20))


def test_backbone_mirror_matches_oracle_on_cpu():
    """ResNetABI_v2_large mirror (PyTorch path, CPU) == the functional oracle == the reference's
    golden outputs for stem + layer1 + layer2."""
    from oracle import tpspp_oracle as TO
    from tps_pp_amd import ResNetABI_v2_large
    m = ResNetABI_v2_large(strides=cases.G7_STRIDES).eval()
    keep = {k: v for k, v in m.state_dict().items() if k.startswith(("conv1.", "bn1.", "layer1.", "layer2."))}
    sd = cases.synth_state(keep, 7, cases.backbone_state_rule)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    got = {}

    class Spy(torch.nn.Module):
        def forward(self, x, outs, **kw):
            got["x"], got["outs"] = x, list(outs)
            return {"output": x}
    img = cases.g7_inputs()["img"]
    with torch.no_grad():
        m._forward_torch(torch.from_numpy(img), Spy())
    ox, oouts = TO.backbone_stem(dict(m.state_dict()), img)
    assert torch.equal(got["x"], ox) and all(torch.equal(a, b) for a, b in zip(got["outs"], oouts))
    G = cases.load("backbone_stem")
    np.testing.assert_allclose(got["x"].numpy(), G["x"], atol=1e-4, rtol=1e-5)
    assert len(m.state_dict()) == 295          # same number of entries as the reference's backbone


def test_nrtr_stem_mirror_layout_and_registry():
    from tps_pp_amd import NRTRModalityTransform
    m = build_backbone(dict(type="NRTRModalityTransform"))
    assert isinstance(m, NRTRModalityTransform)
    keys = list(m.state_dict())
    assert keys[:2] == ["conv_1.weight", "conv_1.bias"] and "bn_2.running_var" in keys and keys[-2:] == ["linear.weight", "linear.bias"]
    with torch.no_grad():
        assert m.eval()._forward_torch(torch.zeros(1, 3, 32, 100)).shape == torch.Size([1, 512, 1, 25])
    with torch.no_grad(), pytest.raises(_lib.TpsppError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 100))


# ---- recogniser head mirrors ----------------------------------------------
def test_head_state_dict_layout_and_registries():
    import json
    import os
    import tps_pp_amd as P
    keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")))
    enc = P.build_encoder(dict(type="NRTREncoder"))
    dec = P.build_decoder(dict(type="NRTRDecoder", num_classes=93, start_idx=91, padding_idx=92, max_seq_len=40))
    assert {k: list(v.shape) for k, v in enc.state_dict().items()} == keys["NRTREncoder"]
    assert {k: list(v.shape) for k, v in dec.state_dict().items()} == keys["NRTRDecoder"]
    # the config of the reference builds end to end (configs/textrecog/nrtr/nrtr_tps++.py:23-42)
    label_convertor = dict(type="AttnConvertor", dict_type="DICT90", with_unknown=True)
    m = P.build_detector(dict(type="NRTR",
                              backbone=dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3],
                                            strides=[2, 1, 2, 1, 2]),
                              tpsnet=dict(type="TPS_PP"), encoder=dict(type="NRTREncoder"),
                              decoder=dict(type="NRTRDecoder"), loss=dict(type="TFLoss"),
                              label_convertor=label_convertor, max_seq_len=40))
    assert m.tpsnet.type == "ResNet45" and not m.tpsnet.variant_explicit      # from the backbone's strides [2,1,...]
    m_v2 = P.build_detector(dict(type="NRTR", backbone=dict(type="ResNetABI_v2_large", arch_settings=[1, 1, 1, 1, 1],
                                                            strides=[1, 2, 2, 1, 2]),
                                 tpsnet=dict(type="TPS_PP"), encoder=dict(type="NRTREncoder", n_layers=1),
                                 decoder=dict(type="NRTRDecoder", n_layers=1), loss=dict(type="TFLoss"),
                                 label_convertor=label_convertor, max_seq_len=40))
    assert m_v2.tpsnet.type == "ResNet45v2"
    assert (m.decoder.start_idx, m.decoder.padding_idx, m.decoder.max_seq_len) == (91, 92, 40)
    assert m.decoder.classifier.out_features == 92 and m.decoder.trg_word_emb.num_embeddings == 93
    with pytest.raises(Exception, match="GPU"):
        m.simple_test(torch.zeros(1, 3, 32, 128), [dict(resize_shape=(32, 128, 3))])
    # the training graph (round 5) needs the GPU as well: its transformation stage is the HIP warp, forward and backward
    with pytest.raises(Exception, match="GPU"):
        m.train().forward_train(torch.zeros(1, 3, 32, 128), [dict(resize_shape=(32, 128, 3), text="ab")])
    from tps_pp_amd import losses
    assert isinstance(m.loss, losses.TFLoss) and m.loss.ignore_index == 92 and m.loss.shift and m.loss.flatten


def test_attn_convertor_matches_the_reference():
    import tps_pp_amd as P
    G = cases.load("nrtr_head_full")
    c = P.build_convertor(dict(type="AttnConvertor", dict_type="DICT90", with_unknown=True, max_seq_len=40))
    assert (c.unknown_idx, c.start_idx, c.end_idx, c.padding_idx, c.num_classes()) == (90, 91, 91, 92, 93)
    assert np.array_equal(c.str2tensor(["hello", "W0rld!"])["padded_targets"].numpy(), G["str2tensor_targets"])
    idx, scores = c.tensor2idx(torch.from_numpy(G["out_dec"]))
    assert c.idx2str(idx) == [str(s) for s in G["text"]] and [len(i) for i in idx] == G["idx_len"].tolist()
    assert all(abs(s - float(G["out_dec"][0, j].max())) < 1e-7 for j, s in enumerate(scores[0]))
    assert c.str2idx(["aé"]) == [[10, 90]]                       # unknown character -> <UKN>
    c36 = P.AttnConvertor(dict_type="DICT36", with_unknown=False, lower=True, start_end_same=False)
    assert (c36.start_idx, c36.end_idx, c36.padding_idx) == (36, 37, 38) and c36.str2idx(["AB"]) == [[10, 11]]
    with pytest.raises(Exception):
        c36.str2idx(["!"])
    # <EOS> ends a string, <PAD> is skipped
    out = torch.zeros(1, 5, 92)
    for t, k in enumerate([10, 11, 91, 12, 13]):
        out[0, t, k] = 1.0
    assert c.idx2str(c.tensor2idx(out)[0]) == ["ab"]
    # tensor2str (what simple_test calls) = idx2str(tensor2idx()) in one pass, multi-character tokens (<UKN>) included
    st, sc = c.tensor2str(torch.from_numpy(G["out_dec"]))
    assert st == [str(s) for s in G["text"]] and sc == scores
    g = torch.Generator().manual_seed(4)
    x = torch.rand((33, 40, 92), generator=g)
    x[:, :, c.end_idx] += (torch.rand((33, 40), generator=g) < 0.05).float()
    x[:, :, c.unknown_idx] += (torch.rand((33, 40), generator=g) < 0.03).float()
    i2, s2 = c.tensor2idx(x)
    assert c.tensor2str(x) == (c.idx2str(i2), s2) and any("<UKN>" in t for t in c.idx2str(i2))


def test_head_rejects_unsupported_configurations():
    import tps_pp_amd as P
    with pytest.raises(NotImplementedError):
        P.NRTREncoder(d_k=32, n_head=16)
    with pytest.raises(NotImplementedError):
        P.NRTREncoder(operation_order=("self_attn", "norm", "ffn", "norm"))
    with pytest.raises(NotImplementedError):
        P.NRTREncoder(act_cfg=dict(type="Relu"))
    with pytest.raises(Exception, match="GPU"):
        P.NRTREncoder(n_layers=1)(torch.zeros(1, 512, 2, 4))


def test_ocr_metric_matches_the_reference():
    import json
    import os
    from tps_pp_amd import metrics
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ocr_metric.json")))
    preds, gts = [p for p, _ in G["pairs"]], [g for _, g in G["pairs"]]
    got = metrics.count_matches(preds, gts)
    want = G["count_matches"]
    assert {k: got[k] for k in want if k != "ned"} == {k: v for k, v in want.items() if k != "ned"}
    assert abs(got["ned"] - want["ned"]) < 1e-12
    assert metrics.eval_ocr_metric(preds, gts) == G["eval_ocr_metric"]
    for (p, g), w in zip(G["pairs"], G["per_pair"]):
        r = metrics.count_matches([p], [g])
        assert all(r[k] == w[k] for k in w if k != "ned") and abs(r["ned"] - w["ned"]) < 1e-12, (p, g)
    full = metrics.eval_ocr_metric(["abc", "Abd"], ["abc", "abd"], all_metrics=True)
    assert full["word_acc"] == 0.5 and full["word_acc_ignore_case"] == 1.0 and full["1-N.E.D"] == 1.0
    assert metrics.levenshtein("kitten", "sitting") == 3 and metrics.levenshtein("", "abc") == 3


def test_recognizer_set_compute_dtype_switches_every_stage():
    import tps_pp_amd as P
    m = P.build_detector(dict(
        type="NRTR", backbone=dict(type="ResNetABI_v2_large", arch_settings=[1, 1, 1, 1, 1], strides=[2, 1, 2, 1, 2]),
        tpsnet=dict(type="TPS_PP", variant="ResNet45"), encoder=dict(type="NRTREncoder", n_layers=1),
        decoder=dict(type="NRTRDecoder", n_layers=1), loss=dict(type="TFLoss"),
        label_convertor=dict(type="AttnConvertor", dict_type="DICT90", with_unknown=True), max_seq_len=40))
    m.set_compute_dtype("bf16x3")
    assert m.backbone.compute_dtype == m.tpsnet.compute_dtype == m.encoder.compute_dtype == m.decoder.compute_dtype == "bf16x3"
    m.set_compute_dtype(torch.bfloat16)
    assert m.backbone.compute_dtype == torch.bfloat16 and m.tpsnet.compute_dtype is None
    assert m.encoder.compute_dtype == torch.bfloat16 and m.decoder.compute_dtype == torch.bfloat16
    m.set_compute_dtype(None)
    assert m.backbone.compute_dtype is None and m.encoder.compute_dtype is None
    import pytest
    with pytest.raises(ValueError):
        m.set_compute_dtype(torch.float16)


def test_attn_convertor_tensor2idx_scan_matches_the_reference_loop():
    """`tensor2idx` does the reference's per-character scan (convertors/attn.py:124-137: skip <PAD>, stop at the first
    <EOS>) as array operations; here against the literal loop."""
    import tps_pp_amd as P
    c = P.AttnConvertor(dict_type="DICT90", with_unknown=True)
    g = torch.Generator().manual_seed(0)
    out = torch.rand((41, 40, 93), generator=g)
    for i in range(41):
        for j in torch.randint(0, 40, (3,), generator=g).tolist():
            out[i, j, c.end_idx if (i + j) % 2 else c.padding_idx] = 2.0
    out[0, :, c.end_idx] = 0.0                                  # a row without <EOS>
    max_value, max_idx = torch.max(out, -1)
    want_i, want_s = [], []
    for row_idx, row_val in zip(max_idx.tolist(), max_value.tolist()):
        si, ss = [], []
        for ci, cs in zip(row_idx, row_val):
            if ci == c.padding_idx:
                continue
            if ci == c.end_idx:
                break
            si.append(ci)
            ss.append(cs)
        want_i.append(si)
        want_s.append(ss)
    got_i, got_s = c.tensor2idx(out)
    assert got_i == want_i and got_s == want_s


def test_head_training_graphs_reproduce_the_reference():
    """Round 5: under `.train()` NRTREncoder / NRTRDecoder run as PyTorch compositions of their own layers
    (`_forward_graph`, `_forward_train_graph`) so that autograd reaches the parameters.  Pinned here, on the CPU, to the
    reference's own outputs (goldens G9 / G10: nrtr_encoder.py:66-87 with and without the valid-ratio mask;
    nrtr_decoder.py:95-151 teacher-forced logits) in eval mode (dropout off); then in train mode with dropout 0 the same
    numbers, and gradients reach every parameter."""
    import tps_pp_amd as P
    cfg = dict(cases.HD_SMALL)

    def load(m, seed):
        sd = cases.synth_state(m.state_dict(), seed, cases.head_state_rule, cases.HD_KEEP)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        return m
    enc = load(P.NRTREncoder(**cfg).eval(), 9)
    dec = load(P.NRTRDecoder(d_embedding=cfg["d_model"], num_classes=cases.NUM_CLASSES, start_idx=cases.START_IDX,
                             padding_idx=cases.PAD_IDX, max_seq_len=cases.HD_MAXLEN, **cfg).eval(), 10)
    GE, GD = cases.load("nrtr_encoder"), cases.load("nrtr_decoder")
    feat = torch.from_numpy(cases.g9_inputs()["feat"])
    metas = [dict(valid_ratio=r) for r in cases.HD_RATIOS]
    inp = cases.g10_inputs()
    out_enc, tgt = torch.from_numpy(inp["out_enc"]), torch.from_numpy(inp["padded_targets"])
    with torch.no_grad():
        assert np.abs(enc._forward_graph(feat, metas).numpy() - GE["out_masked"]).max() <= 2e-6
        assert np.abs(enc._forward_graph(feat, None).numpy() - GE["out_nomask"]).max() <= 2e-6
        assert np.abs(dec._forward_train_graph(out_enc, tgt, metas).numpy() - GD["logits"]).max() <= 2e-5
    # train mode, dropout rate 0: same arithmetic, and a loss reaches every parameter of both modules
    enc.train(), dec.train()
    enc.dropout_p = dec.dropout_p = 0.0
    f = feat.clone().requires_grad_(True)
    e = enc._forward_graph(f, metas)
    logits = dec._forward_train_graph(e, tgt, metas)
    assert np.abs(logits.detach().numpy() - dec._forward_train_graph(enc._forward_graph(feat, metas), tgt, metas).detach().numpy()).max() == 0
    from tps_pp_amd import losses
    loss = losses.TFLoss(ignore_index=cases.PAD_IDX)(logits, dict(padded_targets=tgt))["loss_ce"]
    assert loss.shape == (cases.HD_N * (cases.HD_MAXLEN - 1),) and bool((loss[tgt[:, 1:].reshape(-1) == cases.PAD_IDX] == 0).all())
    loss.sum().backward()
    assert f.grad is not None and float(f.grad.abs().max()) > 0
    for name, prm in list(enc.named_parameters()) + list(dec.named_parameters()):
        assert prm.grad is not None, name
        if "trg_word_emb" not in name:
            assert float(prm.grad.abs().max()) > 0, name
    # dropout on: outputs differ from run to run (the graph really applies it where the reference does)
    enc.dropout_p = 0.5
    with torch.no_grad():
        a, b = enc._forward_graph(feat, metas), enc._forward_graph(feat, metas)
    assert not torch.equal(a, b)
    # losses: CELoss / TFLoss against a direct F.cross_entropy
    import torch.nn.functional as Fn
    lg = torch.randn(2, 5, 7)
    tg = torch.tensor([[6, 1, 2, 3, 0], [6, 4, 0, 0, 0]])
    want = Fn.cross_entropy(lg[:, :-1].reshape(-1, 7), tg[:, 1:].reshape(-1), ignore_index=0, reduction="none")
    assert torch.equal(losses.build_loss(dict(type="TFLoss", ignore_index=0))(lg, dict(padded_targets=tg))["loss_ce"], want)
    ce = losses.build_loss(dict(type="CELoss", ignore_index=0, reduction="mean"))(lg, dict(padded_targets=tg))["loss_ce"]
    assert abs(float(ce) - float(Fn.cross_entropy(lg.permute(0, 2, 1), tg, ignore_index=0))) < 1e-6
    with pytest.raises(AssertionError):
        losses.CELoss(reduction="avg")


def test_default_arithmetic_configuration_follows_the_environment(monkeypatch):
    """`TPSPP_COMPUTE_DTYPE` (tps_pp_amd/precision.py) is read when a module is constructed: unset / fp32 -> the exact fp32
    kernels (the default: the reference's own arithmetic), bf16x3 -> every stage on the three-term split (the parity
    configuration of configs[4], for a deployment that has checked its checkpoint), bf16 -> the throughput configuration
    (TPS_PP then follows its input's dtype); anything else is an error, not a silent fp32."""
    import tps_pp_amd as P
    mk = lambda: (P.TPS_PP(), P.build_backbone(dict(type="ResNetABI_v2_large", arch_settings=[1, 1, 1, 1, 1], strides=[2, 1, 2, 1, 2])),  # noqa: E731
                  P.NRTREncoder(n_layers=1), P.NRTRDecoder(n_layers=1, num_classes=93, max_seq_len=8, start_idx=91, padding_idx=92))
    monkeypatch.delenv("TPSPP_COMPUTE_DTYPE", raising=False)
    assert [m.compute_dtype for m in mk()] == [None] * 4
    monkeypatch.setenv("TPSPP_COMPUTE_DTYPE", "bf16x3")
    assert [m.compute_dtype for m in mk()] == ["bf16x3"] * 4
    monkeypatch.setenv("TPSPP_COMPUTE_DTYPE", "bf16")
    assert [m.compute_dtype for m in mk()] == [None, torch.bfloat16, torch.bfloat16, torch.bfloat16]
    monkeypatch.setenv("TPSPP_COMPUTE_DTYPE", "fp16")
    with pytest.raises(ValueError, match="TPSPP_COMPUTE_DTYPE"):
        P.NRTREncoder(n_layers=1)
