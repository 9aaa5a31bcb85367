"""-m gpu: the recogniser head (NRTR encoder / decoder / convertor) on a real
MI355X against the reference's golden outputs and the CPU oracle.

Floating-point transformer arithmetic on the fp32 matrix cores: the bar is the north-star tolerance
1e-4 on activations / scores, and EXACT agreement of the decoded token indices and strings."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from oracle import nrtr_oracle as NO
from tps_pp_amd import AttnConvertor, NRTRDecoder, NRTREncoder, ops

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def load_synth(m, seed):
    sd = cases.synth_state(m.state_dict(), seed, cases.head_state_rule, cases.HD_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m


def small_modules(cuda):
    cfg = dict(cases.HD_SMALL)
    enc = load_synth(NRTREncoder(**cfg).eval(), 9).to(cuda)
    dec = load_synth(NRTRDecoder(d_embedding=cfg["d_model"], num_classes=cases.NUM_CLASSES,
                                 start_idx=cases.START_IDX, padding_idx=cases.PAD_IDX,
                                 max_seq_len=cases.HD_MAXLEN, **cfg).eval(), 10).to(cuda)
    return enc, dec


def full_modules(cuda):
    enc = load_synth(NRTREncoder().eval(), 9).to(cuda)
    dec = load_synth(NRTRDecoder(num_classes=cases.NUM_CLASSES, start_idx=cases.START_IDX,
                                 padding_idx=cases.PAD_IDX, max_seq_len=40).eval(), 10).to(cuda)
    return enc, dec


def test_layernorm_transpose_and_gelu_kernels(cuda):
    g = torch.Generator().manual_seed(3)
    for C, M in [(128, 60), (512, 1000), (96, 64)]:
        x = torch.randn(C, M, generator=g)
        ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
        ref = F.layer_norm(x.t(), (C,), ga, be, 1e-5).t()
        got = ops.layernorm_cm(x.to(cuda), ga.to(cuda), be.to(cuda), 1e-5).cpu()
        assert (got - ref).abs().max() < 2e-5
        assert torch.equal(ops.transpose2d(x.to(cuda)).cpu(), x.t().contiguous())
    # GELU (erf form) + bias + residual in the projection epilogue
    K, Co, M = 128, 96, 200
    w, b = torch.randn(Co, K, generator=g) / K ** 0.5, torch.randn(Co, generator=g)
    x, r = torch.randn(K, M, generator=g), torch.randn(Co, M, generator=g)
    cw = ops.prep_conv_weight(w.view(Co, K, 1, 1).to(cuda), conv_bias=b.to(cuda))
    got = ops.conv2d([x.to(cuda).view(1, K, 1, M)], cw, 1, relu=2, residual=r.to(cuda).view(1, Co, 1, M),
                     res_mode=1).view(Co, M).cpu()
    ref = F.gelu(w @ x + b[:, None]) + r
    assert (got - ref).abs().max() < 2e-5


@pytest.mark.parametrize("K,M,Co", [(512, 512, 512), (512, 77, 92), (128, 33, 384), (96, 200, 40)])
def test_fused_layernorm_projection(cuda, K, M, Co):
    g = torch.Generator().manual_seed(K + M)
    x = torch.randn(K, M, generator=g) * 2 + 0.5
    ga, be = torch.randn(K, generator=g), torch.randn(K, generator=g)
    w, b = torch.randn(Co, K, generator=g) / K ** 0.5, torch.randn(Co, generator=g)
    r = torch.randn(Co, M, generator=g)
    ln = F.layer_norm(x.t(), (K,), ga, be, 1e-5)                       # (M, K)
    d = lambda t: t.to(cuda)  # noqa: E731
    folded = ops.fold_layernorm(d(ga), d(be), ops.kmajor(d(w)), d(b))
    got = ops.linear_ln(d(x), folded, 1e-5, act=2, residual=d(r)).cpu()
    assert (got - (F.gelu(ln @ w.t() + b).t() + r)).abs().max() < 5e-5
    folded = ops.fold_layernorm(d(ga), d(be), ops.kmajor(d(w)))
    got_t = ops.linear_ln(d(x), folded, 1e-5, token_major=True).cpu()
    assert (got_t - ln @ w.t()).abs().max() < 5e-5


@pytest.mark.parametrize("T", [20, 64, 100, 256])
def test_encoder_attention_kernel(cuda, T):
    g = torch.Generator().manual_seed(T)
    N, H = 3, 2
    C = 64 * H
    qkv = torch.randn(3 * C, N * T, generator=g)
    vl = torch.tensor([T, max(1, T // 2), max(1, T // 3)], dtype=torch.int32)
    got = ops.attn_enc(qkv.to(cuda), N, T, vl.to(cuda)).cpu()
    got_nomask = ops.attn_enc(qkv.to(cuda), N, T, None).cpu()

    def ref(mask_len):
        q, k, v = (qkv[i * C:(i + 1) * C].view(H, 64, N, T).permute(2, 0, 3, 1) for i in range(3))  # N,H,T,64
        a = torch.matmul(q / 8.0, k.transpose(2, 3))
        if mask_len is not None:
            m = torch.arange(T)[None, :] < mask_len[:, None].long()
            a = a.masked_fill(~m[:, None, None, :], float("-inf"))
        o = torch.matmul(F.softmax(a, -1), v)                      # N,H,T,64
        return o.permute(1, 3, 0, 2).reshape(C, N * T)
    assert (got - ref(vl)).abs().max() < 2e-5
    assert (got_nomask - ref(None)).abs().max() < 2e-5


def test_encoder_small_against_reference(cuda):
    G = cases.load("nrtr_encoder")
    enc, _ = small_modules(cuda)
    feat = dev(cases.g9_inputs()["feat"], cuda)
    metas = [dict(valid_ratio=r) for r in cases.HD_RATIOS]
    with torch.no_grad():
        out_m = enc(feat, metas)
        out_n = enc(feat, None)
    assert out_m.shape == (cases.HD_N, cases.HD_HW[0] * cases.HD_HW[1], cases.HD_SMALL["d_model"])
    assert np.abs(out_m.cpu().numpy() - G["out_masked"]).max() <= TOL
    assert np.abs(out_n.cpu().numpy() - G["out_nomask"]).max() <= TOL
    # the channel-major side output is the same tensor, re-laid out
    cm = out_n._tpspp_cm.cpu().numpy()
    assert np.array_equal(cm.T.reshape(out_n.shape), out_n.cpu().numpy())


def test_decoder_small_against_reference(cuda):
    G = cases.load("nrtr_decoder")
    _, dec = small_modules(cuda)
    inp = cases.g10_inputs()
    out_enc = dev(inp["out_enc"], cuda)
    metas = [dict(valid_ratio=r) for r in cases.HD_RATIOS]
    with torch.no_grad():
        logits = dec(None, out_enc, dict(padded_targets=torch.from_numpy(inp["padded_targets"])), metas,
                     train_mode=True)
        probs = dec(None, out_enc, None, metas, train_mode=False)
        tokens = dec.last_tokens.cpu().numpy()
        probs_nm = dec(None, out_enc, None, None, train_mode=False)
    assert np.abs(logits.cpu().numpy() - G["logits"]).max() <= TOL
    assert np.abs(probs.cpu().numpy() - G["probs"]).max() <= TOL
    assert np.abs(probs_nm.cpu().numpy() - G["probs_nomask"]).max() <= TOL
    assert np.array_equal(tokens[:, 1:], G["probs"].argmax(-1))
    assert (tokens[:, 0] == cases.START_IDX).all()


def test_head_full_size_against_reference(cuda):
    G = cases.load("nrtr_head_full")
    enc, dec = full_modules(cuda)
    conv = AttnConvertor(dict_type="DICT90", with_unknown=True)
    feat = dev(cases.g11_inputs()["feat"], cuda)
    with torch.no_grad():
        out_enc = enc(feat, None)
        out_dec = dec(None, out_enc, None, None, train_mode=False)
    assert np.abs(out_enc.cpu().numpy()[:, :, ::8] - G["out_enc_sub"]).max() <= TOL
    assert np.abs(out_dec.cpu().numpy() - G["out_dec"]).max() <= TOL
    assert np.array_equal(out_dec.argmax(-1).cpu().numpy(), G["argmax"])
    idx, scores = conv.tensor2idx(out_dec)
    assert conv.idx2str(idx) == [str(s) for s in G["text"]]
    assert [len(i) for i in idx] == G["idx_len"].tolist()


def test_head_batch_against_oracle(cuda):
    """A batch with ragged valid ratios, reference-size head: decoded strings identical to the oracle's
    (which re-runs the padded sequence every step, as the reference does)."""
    from tps_pp_amd import synth
    enc, dec = full_modules(cuda)
    n = 12
    feat = synth.dyadic((n, 512, 4, 16), "head.batch", 5)
    ratios = [1.0, 0.9, 0.75, 0.5, 0.3, 1.0, 0.62, 0.11, 1.0, 0.8, 0.45, 0.97]
    metas = [dict(valid_ratio=r) for r in ratios]
    with torch.no_grad():
        out_enc = enc(dev(feat, cuda), metas)
        out_dec = dec(None, out_enc, None, metas, train_mode=False)
    enc_sd = {k: v.cpu() for k, v in enc.state_dict().items()}
    dec_sd = {k: v.cpu() for k, v in dec.state_dict().items()}
    o = NO.head_simple_test(enc_sd, dec_sd, feat, valid_ratios=ratios)
    assert (out_enc.cpu() - o["out_enc"]).abs().max() <= TOL
    assert (out_dec.cpu() - o["out_dec"]).abs().max() <= TOL
    conv = AttnConvertor()
    idx, _ = conv.tensor2idx(out_dec)
    assert idx == o["indexes"] and conv.idx2str(idx) == o["text"]


# The model dict of configs/textrecog/nrtr/nrtr_tps++.py:26-42, values typed here (its `label_convertor` comes from
# _base_ with a dictionary file: DICT90 + <UKN> is the same 93-class table).  No `variant` for TPS_PP: the recogniser
# picks the wiring that fits its backbone's strides, so the config runs unchanged.
NRTR_TPSPP_CONFIG_MODEL = dict(
    type="NRTR",
    backbone=dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3], strides=[2, 1, 2, 1, 2]),
    tpsnet=dict(type="TPS_PP"),
    encoder=dict(type="NRTREncoder"),
    decoder=dict(type="NRTRDecoder"),
    loss=dict(type="TFLoss"),
    label_convertor=dict(type="AttnConvertor", dict_type="DICT90", with_unknown=True),
    max_seq_len=40)


def build_recognizer(cuda):
    """configs/textrecog/nrtr/nrtr_tps++.py:26-42, exactly as written there."""
    import tps_pp_amd as P
    assert NRTR_TPSPP_CONFIG_MODEL["backbone"]["strides"] == list(cases.G12_STRIDES)
    m = P.build_detector(NRTR_TPSPP_CONFIG_MODEL).eval()
    assert m.tpsnet.type == "ResNet45" and not m.tpsnet.variant_explicit
    for mod, seed, rule, keep in ((m.backbone, 7, cases.backbone_state_rule, ()),
                                  (m.tpsnet, 4, cases.tpspp_state_rule, cases.TPSPP_KEEP),
                                  (m.encoder, 9, cases.head_state_rule, cases.HD_KEEP),
                                  (m.decoder, 10, cases.head_state_rule, cases.HD_KEEP)):
        sd = cases.synth_state(mod.state_dict(), seed, rule, keep)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.to(cuda)


def test_recognizer_end_to_end_against_reference(cuda):
    """Image batch -> strings through every HIP stage (backbone convs, TPS++ regressor + warp, encoder,
    greedy decoder, convertor) against the reference's own composition of its modules."""
    G = cases.load("recognizer_e2e")
    m = build_recognizer(cuda)
    img = dev(cases.g12_inputs()["img"], cuda)
    metas = [dict(resize_shape=(32, w, 3)) for w in cases.G12_WIDTHS]
    with torch.no_grad():
        res = m(img, metas, return_loss=False)
        feat = m.extract_feat(img, test=True)["output"]
    assert np.abs(feat.cpu().numpy()[:, ::8] - G["feat_sub"]).max() <= TOL
    assert [r["text"] for r in res] == [str(s) for s in G["text"]]
    assert np.abs(np.array(res[0]["score"], dtype=np.float32) - G["score0"]).max() <= TOL
    assert [m_["valid_ratio"] for m_ in metas] == [w / 128 for w in cases.G12_WIDTHS]


def test_recognizer_bf16_backbone_against_reference(cuda):
    """BASELINE.json configs[4]: the recogniser with backbone + TPS++ convolutions on the bf16 matrix cores
    (`backbone.compute_dtype = torch.bfloat16`; the transformer head stays fp32) against the reference's fp32
    run (golden G12): the feature map within bf16 resolution, the decoded strings identical."""
    G = cases.load("recognizer_e2e")
    m = build_recognizer(cuda)
    m.backbone.compute_dtype = torch.bfloat16
    img = dev(cases.g12_inputs()["img"], cuda)
    metas = [dict(resize_shape=(32, w, 3)) for w in cases.G12_WIDTHS]
    with torch.no_grad():
        res = m(img, metas, return_loss=False)
        feat = m.extract_feat(img, test=True)["output"]
    assert feat.dtype == torch.float32
    ref = G["feat_sub"]
    err = np.abs(feat.cpu().numpy()[:, ::8] - ref)
    assert err.max() <= 0.05 * np.abs(ref).max() and err.mean() <= 0.005 * np.abs(ref).max()
    assert [r["text"] for r in res] == [str(s) for s in G["text"]]
    # ... and with the head's wide projections on the bf16 matrix cores and bf16 encoder keys / values
    # (TPSPP_HEAD_BF16): same strings, per-character scores at bf16 resolution
    m.encoder.compute_dtype = m.decoder.compute_dtype = torch.bfloat16
    with torch.no_grad():
        res16 = m(img, [dict(mm) for mm in metas], return_loss=False)
    assert [r["text"] for r in res16] == [str(s) for s in G["text"]]
    assert np.abs(np.array(res16[0]["score"], dtype=np.float32) - G["score0"]).max() <= 2e-2


def test_recognizer_bf16x3_against_reference(cuda):
    """The "bf16x3" configuration end to end (fp32 tensors, three-term bf16 split in the backbone / TPS++ convolutions
    and the head's wide projections): the north-star's 1e-4 bar against the reference's fp32 run (golden G12)."""
    G = cases.load("recognizer_e2e")
    m = build_recognizer(cuda)
    m.backbone.compute_dtype = m.tpsnet.compute_dtype = m.encoder.compute_dtype = m.decoder.compute_dtype = "bf16x3"
    img = dev(cases.g12_inputs()["img"], cuda)
    metas = [dict(resize_shape=(32, w, 3)) for w in cases.G12_WIDTHS]
    with torch.no_grad():
        res = m(img, metas, return_loss=False)
        feat = m.extract_feat(img, test=True)["output"]
    assert np.abs(feat.cpu().numpy()[:, ::8] - G["feat_sub"]).max() <= TOL
    assert [r["text"] for r in res] == [str(s) for s in G["text"]]
    assert np.abs(np.array(res[0]["score"], dtype=np.float32) - G["score0"]).max() <= TOL


def test_head_bf16_flag_against_fp32_head(cuda):
    """TPSPP_HEAD_BF16 on the small head (odd sizes, key masks): encoder output and decoder probabilities against
    the fp32 HIP head at bf16 resolution; greedy tokens identical."""
    enc, dec = small_modules(cuda)
    feat = dev(cases.g9_inputs()["feat"], cuda)
    metas = [dict(valid_ratio=r) for r in cases.HD_RATIOS]
    with torch.no_grad():
        e32 = enc(feat, metas)
        p32 = dec(feat, e32, None, metas, train_mode=False)
        t32 = dec.last_tokens.clone()
        enc.compute_dtype = dec.compute_dtype = torch.bfloat16
        e16 = enc(feat, metas)
        p16 = dec(feat, e16, None, metas, train_mode=False)
        t16 = dec.last_tokens.clone()
    assert (e16 - e32).abs().max().item() <= 3e-2 * e32.abs().max().item()
    assert (p16 - p32).abs().max().item() <= 2e-2
    assert torch.equal(t16, t32)


@pytest.mark.parametrize("hw,n", [((4, 20), 5), ((1, 7), 3), ((4, 40), 2)])
def test_head_other_token_counts_against_oracle(cuda, hw, n):
    """Token counts other than 64 (wider / narrower images: T = 80, 7, 160 > one wavefront), odd batch."""
    from tps_pp_amd import synth
    enc, dec = small_modules(cuda)
    feat = synth.dyadic((n, cases.HD_SMALL["d_model"]) + hw, f"head.T{hw}", 6)
    ratios = [1.0, 0.37, 0.81, 0.5, 0.95][:n]
    metas = [dict(valid_ratio=r) for r in ratios]
    with torch.no_grad():
        out_enc = enc(dev(feat, cuda), metas)
        out_dec = dec(None, out_enc, None, metas, train_mode=False)
    enc_sd = {k: v.cpu() for k, v in enc.state_dict().items()}
    dec_sd = {k: v.cpu() for k, v in dec.state_dict().items()}
    nh = cases.HD_SMALL["n_head"]
    o_enc = NO.encoder_forward(enc_sd, feat, nh, ratios)
    o_dec = NO.decoder_forward_test(dec_sd, o_enc, nh, cases.HD_MAXLEN, cases.START_IDX, cases.PAD_IDX, ratios)
    assert (out_enc.cpu() - o_enc).abs().max() <= TOL
    assert (out_dec.cpu() - o_dec).abs().max() <= TOL
    assert torch.equal(out_dec.argmax(-1).cpu(), o_dec.argmax(-1))


def test_head_argument_errors(cuda):
    from tps_pp_amd import _lib
    enc, dec = small_modules(cuda)
    with torch.no_grad():
        with pytest.raises(_lib.TpsppError, match="256"):
            enc(torch.zeros(1, cases.HD_SMALL["d_model"], 4, 65, device=cuda))       # 260 tokens
        with pytest.raises(ValueError):
            enc(torch.zeros(1, 64, 2, 4, device=cuda))                                # wrong width
        with pytest.raises(ValueError):
            enc(torch.zeros(2, cases.HD_SMALL["d_model"], 2, 4, device=cuda), [dict(valid_ratio=1.0)])
        # the decoder's pointer table travels with its length (ABI version 2): a table in an earlier round's 18-pointer-
        # per-layer layout is refused with TPSPP_EINVAL instead of being read out of bounds
        feat = torch.zeros(2, cases.HD_SMALL["d_model"], 2, 4, device=cuda)
        out_enc = enc(feat, None)
        table = dec._weights()[0]
        good = len(table)
        try:
            table.keep = table.keep[:good - 6]          # what len(table) reports to the C entry point
            with pytest.raises(_lib.TpsppError, match="layer_ptrs_len"):
                dec(feat, out_enc, None, None, train_mode=False)
        finally:
            dec._w_cache = None                          # rebuild the table for whoever uses the module next


def test_decoder_fused_q_cross_launch_is_bit_identical(cuda, tmp_path):
    """TPSPP_HEAD_QCROSS=1 (q projection + cross-attention of a layer-step in one launch; opt-in, measured slower) must
    give exactly the scores of the two-launch path: the switch is read when the library loads, so two processes."""
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "from tps_pp_amd.nrtr_head import NRTRDecoder\n"
        "torch.manual_seed(3)\n"
        "dec = NRTRDecoder(num_classes=93, max_seq_len=6, start_idx=91, padding_idx=92).eval().cuda()\n"
        "enc = torch.randn(37, 64, 512, device='cuda'); feat = torch.empty(37, 512, 8, 8, device='cuda')\n"
        "res = {}\n"
        "for tag, cd in (('bf16x3', 'bf16x3'), ('bf16', torch.bfloat16)):\n"
        "    dec.compute_dtype = cd\n"
        "    with torch.no_grad():\n"
        "        res[tag] = dec(feat, enc, None, None, train_mode=False).cpu()\n"
        "torch.save(res, sys.argv[1])\n")
    import os
    outs = []
    for val in (None, "1"):
        env = dict(os.environ)
        env["TPSPP_HEAD_NO_PERSIST"] = "1"               # both on the launch-per-phase pipeline (the fused launch is one of its phases)
        env.pop("TPSPP_HEAD_QCROSS", None)
        if val:
            env["TPSPP_HEAD_QCROSS"] = val
        path = str(tmp_path / f"dec_{val}.pt")
        subprocess.run([sys.executable, "-c", code, path], env=env, check=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        outs.append(torch.load(path))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
        assert torch.isfinite(outs[0][k]).all()


@pytest.mark.parametrize("n,seq", [(37, 6), (64, 40), (557, 5)])
def test_decoder_persistent_step_matches_the_launch_pipeline(cuda, n, seq):
    """Round 5: a decoding step is ONE persistent launch (tpspp_head_persist.h: clusters of 16
    workgroups per 32 images, cluster barriers, system-scope hand-offs) instead of ~50 dependent launches
    (TPSPP_HEAD_NO_PERSIST=1 selects those; read per call).  Exact-fp32 head: the same arithmetic in the same order, scores
    BIT-IDENTICAL.  Reduced-precision heads: same phases, same per-element arithmetic except that a projection's K is split
    over 8 wavefronts instead of 4: soft-max scores within 5e-5 (the classifier is spread x6 here), decided tokens IDENTICAL -- a stale or torn
    hand-off would show as an O(1) difference.  Greedy and teacher-forced, ragged valid ratios, a batch that is not a
    multiple of the 32-image cluster (37), one of several 512-image launches per step (557), full length (40 steps);
    bf16x3 and bf16 heads; three repetitions (the hand-offs race differently every time)."""
    import os
    from tps_pp_amd.nrtr_head import NRTRDecoder
    torch.manual_seed(5)
    dec = NRTRDecoder(num_classes=93, max_seq_len=seq, start_idx=91, padding_idx=92).eval().to(cuda)
    with torch.no_grad():
        dec.classifier.weight.mul_(6.0)                   # decided tokens far from ties
    enc = torch.randn(n, 64, 512, device=cuda)
    metas = [dict(valid_ratio=(1.0, 0.7, 0.4)[i % 3]) for i in range(n)]
    forced = torch.randint(0, 90, (n, seq), device=cuda)
    forced[:, 0] = 91
    forced[::3, seq - 1] = 92                              # a <PAD> key in some rows
    old = os.environ.pop("TPSPP_HEAD_NO_PERSIST", None)
    try:
        for cd in (None, "bf16x3", torch.bfloat16):
            dec.compute_dtype = cd
            with torch.no_grad():
                os.environ["TPSPP_HEAD_NO_PERSIST"] = "1"
                want = dec(None, enc, None, metas, train_mode=False)
                want_tok = dec.last_tokens.clone()
                want_tf = dec(None, enc, dict(padded_targets=forced), metas, train_mode=True)
                del os.environ["TPSPP_HEAD_NO_PERSIST"]
                for rep in range(3):
                    got = dec(None, enc, None, metas, train_mode=False)
                    got_tok = dec.last_tokens.clone()
                    got_tf = dec(None, enc, dict(padded_targets=forced), metas, train_mode=True)
                    assert torch.isfinite(got).all() and torch.isfinite(got_tf).all(), (cd, rep)
                    bad = (got_tok != want_tok)
                    assert not bool(bad.any()), (cd, rep, int(bad.sum()), bad.nonzero()[:4].tolist(), float((got - want).abs().max()))
                    # (bf16 head: the cached keys / values are rounded to bf16, so last-bit differences of a projection reach 2e-4)
                    if cd is None:      # exact-fp32 head: the same arithmetic in the same order -> the same bits
                        assert torch.equal(got, want) and torch.equal(got_tf, want_tf), (cd, rep, float((got - want).abs().max()))
                    assert float((got - want).abs().max()) <= (5e-5 if cd == "bf16x3" else 5e-4), (cd, rep, float((got - want).abs().max()))
                    assert float((got_tf - want_tf).abs().max()) <= (1e-3 if cd == torch.bfloat16 else 1e-4) * float(want_tf.abs().max()), (cd, rep)
                # the step loop is INSIDE the launch (one launch per decode); one launch per step (TPSPP_HEAD_STEP_LAUNCHES=1) and
                # write-through stores whatever the clusters' placement (TPSPP_HEAD_WRITE_THROUGH=1): the same bits
                for var in ("TPSPP_HEAD_STEP_LAUNCHES", "TPSPP_HEAD_WRITE_THROUGH"):
                    os.environ[var] = "1"
                    one = dec(None, enc, None, metas, train_mode=False)
                    one_tf = dec(None, enc, dict(padded_targets=forced), metas, train_mode=True)
                    del os.environ[var]
                    assert torch.equal(one, got) and torch.equal(one_tf, got_tf), (cd, var, float((one - got).abs().max()))
    finally:
        os.environ.pop("TPSPP_HEAD_NO_PERSIST", None)
        os.environ.pop("TPSPP_HEAD_STEP_LAUNCHES", None)
        os.environ.pop("TPSPP_HEAD_WRITE_THROUGH", None)
        if old is not None:
            os.environ["TPSPP_HEAD_NO_PERSIST"] = old


def test_decoder_persistent_kernel_other_depths_and_widths(cuda):
    """The persistent decoder kernel outside the reference's default shape: an ODD number of layers (the activations' x / y
    buffers end a step swapped: the step loop inside the launch and the host's per-step launches must both follow) and
    d_inner = 512 (four 16-wide k-steps per wavefront in the w2 projection instead of two), 96 and 40 encoder tokens (wider images:
    the cross-attention then walks a wavefront's two heads one after the other), every head configuration, greedy and
    teacher-forced, against the launch pipeline: exact-fp32 head bit-identical, reduced heads within 5e-5 / 5e-4 with identical
    tokens; one launch per step and write-through stores bit-equal to the default."""
    import os
    from tps_pp_amd.nrtr_head import NRTRDecoder
    torch.manual_seed(11)
    n, seq = 45, 7
    for n_layers, d_inner, T in ((3, 512, 64), (1, 256, 64), (2, 512, 64), (2, 256, 96), (1, 256, 40)):
        dec = NRTRDecoder(n_layers=n_layers, d_inner=d_inner, num_classes=93, max_seq_len=seq, start_idx=91, padding_idx=92).eval().to(cuda)
        with torch.no_grad():
            dec.classifier.weight.mul_(6.0)
        enc = torch.randn(n, T, 512, device=cuda)         # (T > 64 encoder tokens: the cross-attention's one-head-at-a-time form)
        metas = [dict(valid_ratio=(1.0, 0.6)[i % 2]) for i in range(n)]
        forced = torch.randint(0, 90, (n, seq), device=cuda)
        forced[:, 0] = 91
        try:
            for cd in (None, "bf16x3", torch.bfloat16):
                dec.compute_dtype = cd
                with torch.no_grad():
                    os.environ["TPSPP_HEAD_NO_PERSIST"] = "1"
                    want = dec(None, enc, None, metas, train_mode=False)
                    want_tok = dec.last_tokens.clone()
                    want_tf = dec(None, enc, dict(padded_targets=forced), metas, train_mode=True)
                    del os.environ["TPSPP_HEAD_NO_PERSIST"]
                    got = dec(None, enc, None, metas, train_mode=False)
                    got_tok = dec.last_tokens.clone()
                    got_tf = dec(None, enc, dict(padded_targets=forced), metas, train_mode=True)
                    tag = (n_layers, d_inner, T, cd)
                    assert torch.isfinite(got).all() and torch.equal(got_tok, want_tok), tag
                    if cd is None:
                        assert torch.equal(got, want) and torch.equal(got_tf, want_tf), tag
                    assert float((got - want).abs().max()) <= (5e-5 if cd != torch.bfloat16 else 5e-4), (tag, float((got - want).abs().max()))
                    assert float((got_tf - want_tf).abs().max()) <= (1e-3 if cd == torch.bfloat16 else 1e-4) * float(want_tf.abs().max()), tag
                    for var in ("TPSPP_HEAD_STEP_LAUNCHES", "TPSPP_HEAD_WRITE_THROUGH"):
                        os.environ[var] = "1"
                        one = dec(None, enc, None, metas, train_mode=False)
                        del os.environ[var]
                        assert torch.equal(one, got), (tag, var)
        finally:
            for var in ("TPSPP_HEAD_NO_PERSIST", "TPSPP_HEAD_STEP_LAUNCHES", "TPSPP_HEAD_WRITE_THROUGH"):
                os.environ.pop(var, None)


TOKGEMM_CASES = [
    # name, K, Co, M, act, residual, out dtype, x3
    ("qkv_wide", 512, 1536, 32768, None, False, torch.float32, False),          # 256-output workgroups
    ("qkv_wide_x3", 512, 1536, 32768, None, False, torch.float32, True),
    ("fc_res", 512, 512, 4096, None, True, torch.float32, False),
    ("w1_gelu", 512, 256, 4096, "gelu", False, torch.float32, False),
    ("w2_res_tail", 256, 512, 1000, None, True, torch.float32, False),          # M not a multiple of the 128-token tile
    ("kv_bf16_out", 512, 512, 2052, None, False, torch.bfloat16, False),
    ("x3_gelu_res_tail", 256, 128, 516, "gelu", True, torch.float32, True),
]


@pytest.mark.parametrize("name,K,Co,M,act,res,odt,x3", TOKGEMM_CASES, ids=[c[0] for c in TOKGEMM_CASES])
def test_token_gemm_against_float64(cuda, name, K, Co, M, act, res, odt, x3):
    """tpspp_token_gemm_bf16_fwd (the head's channel-major Linear on the bf16 matrix cores): against float64 on operands
    rounded the way the kernel rounds them (bf16; x3: fp32 operands, ~5e-6 of the scale); wide and narrow workgroups, a
    tail tile, GELU, residual, bf16 output."""
    g = torch.Generator(device="cpu").manual_seed(K * 7 + Co + M)
    w = (torch.randn((Co, K, 1, 1), generator=g) * 0.05)
    b = torch.randn(Co, generator=g) * 0.1
    x = torch.randn((K, M), generator=g)
    r = torch.randn((Co, M), generator=g) if res else None
    cw = ops.prep_conv_weight_bf16(w.to(cuda), conv_bias=b.to(cuda), x3=x3)
    got = ops.token_gemm_bf16(x.to(cuda), cw, act=act, residual=None if r is None else r.to(cuda), out_dtype=odt)
    assert got.shape == (Co, M) and got.dtype == odt
    if x3:
        wq, xq = w.view(Co, K).double(), x.double()
    else:
        wq, xq = w.view(Co, K).bfloat16().double(), x.bfloat16().double()
    want = wq @ xq + b.double()[:, None]
    if act == "gelu":
        want = torch.nn.functional.gelu(want)
    if r is not None:
        want = want + r.double()
    scale = float(want.abs().max())
    err = float((got.double().cpu() - want).abs().max()) / scale
    tol = 2e-5 if x3 else (6e-3 if odt == torch.bfloat16 else 2e-6)
    assert err < tol, (name, err)


def test_token_gemm_argument_errors(cuda):
    cw = ops.prep_conv_weight_bf16(torch.zeros((128, 64, 1, 1), device=cuda), conv_bias=torch.zeros(128, device=cuda))
    with pytest.raises(Exception):
        ops.token_gemm_bf16(torch.zeros((64, 6), device=cuda), cw)                 # M not a multiple of 4
    with pytest.raises(ValueError):
        ops.token_gemm_bf16(torch.zeros((32, 8), device=cuda), cw)                 # K != Cin
    cw96 = ops.prep_conv_weight_bf16(torch.zeros((96, 64, 1, 1), device=cuda), conv_bias=torch.zeros(96, device=cuda))
    with pytest.raises(Exception):
        ops.token_gemm_bf16(torch.zeros((64, 8), device=cuda), cw96)               # Co not a multiple of 128


def test_recognizer_training_steps_through_the_hip_warp(cuda):
    """Round 5 (README.md:60-64 of the reference trains nrtr_tps++.py through mmocr/apis/train.py:56-70):
    the recogniser built from the config's model dict, `.train()`, `forward(img, metas, return_loss=True)` -> the loss dict
    of TFLoss.  Every stage is the PyTorch composition of its own layers; the TPS++ transformation stage inside the
    backbone runs on the HIP kernels in both directions (ops.warp_autograd).  Gradients reach the parameters of all four
    modules -- those of the TPS++ regressor only THROUGH tpspp_warp_bwd -- and a small enough SGD step lowers the loss."""
    torch.manual_seed(3)
    m = build_recognizer(cuda).train()
    m.encoder.dropout_p = m.decoder.dropout_p = 0.0                # deterministic steps
    n = len(cases.G12_WIDTHS)
    img = dev(cases.g12_inputs()["img"], cuda)
    texts = ["hello", "W0rld!", "tps++", "a"][:n] + ["x"] * max(0, n - 4)
    metas = [dict(resize_shape=(32, w, 3), text=t) for w, t in zip(cases.G12_WIDTHS, texts)]
    calls = []
    orig = ops.warp_backward

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    ops.warp_backward = spy
    try:
        out = m(img, [dict(mm) for mm in metas], return_loss=True)
        assert set(out) == {"loss_ce"} and out["loss_ce"].shape == (n * 39,)
        first = out["loss_ce"].sum() / max(1, int((out["loss_ce"] != 0).sum()))
        first.backward()
    finally:
        ops.warp_backward = orig
    assert calls, "the transformation stage's backward did not run on the HIP kernel"
    probes = {"backbone stem": m.backbone.conv1.weight, "backbone layer5": m.backbone.layer5[0].conv2.weight,
              "TPS++ control points (only reachable through the warp's backward)": m.tpsnet.TPE.localization_fc2.bias,
              "TPS++ score": m.tpsnet.TPE.feat_linear[0].weight, "TPS++ down2": m.tpsnet.down2.conv.weight,
              "encoder": m.encoder.layer_stack[0].attn.linear_q.weight, "decoder": m.decoder.layer_stack[5].mlp.w_2.weight,
              "classifier": m.decoder.classifier.weight}
    for name, prm in probes.items():
        assert prm.grad is not None and torch.isfinite(prm.grad).all() and float(prm.grad.abs().max()) > 0, name
    # the gradient is a descent direction: from the same start, a small enough SGD step lowers the loss
    def mean_loss():
        o = m(img, [dict(mm) for mm in metas], return_loss=True)["loss_ce"]
        return o.sum() / max(1, int((o != 0).sum()))
    start = {k: v.clone() for k, v in m.state_dict().items()}
    tried = {}
    for lr in (1e-2, 1e-3, 1e-4, 1e-5):
        m.load_state_dict(start)
        opt = torch.optim.SGD(m.parameters(), lr=lr)
        opt.zero_grad()
        l0 = mean_loss()
        l0.backward()
        opt.step()
        with torch.no_grad():
            tried[lr] = (float(l0.detach()), float(mean_loss().detach()))
        if np.isfinite(tried[lr][1]) and tried[lr][1] < tried[lr][0]:
            break
    assert any(np.isfinite(b) and b < a for a, b in tried.values()), tried
    m.load_state_dict(start)
    # and inference still runs on the HIP kernels afterwards
    m.eval()
    with torch.no_grad():
        res = m(img, [dict(mm) for mm in metas], return_loss=False)
    assert len(res) == n and all(isinstance(r["text"], str) for r in res)


# ---- round 6: the persistent decode is safe to overlap, and its failure is loud -----------------------------------------------
def _decoders(cuda, count, seq=40):
    """`count` NRTRDecoder modules with the same weights (each has its own workspace, as independent requests have)."""
    from tps_pp_amd.nrtr_head import NRTRDecoder
    torch.manual_seed(21)
    first = NRTRDecoder(num_classes=93, max_seq_len=seq, start_idx=91, padding_idx=92).eval().to(cuda)
    with torch.no_grad():
        first.classifier.weight.mul_(6.0)
    decs = [first]
    for _ in range(count - 1):
        d = NRTRDecoder(num_classes=93, max_seq_len=seq, start_idx=91, padding_idx=92).eval().to(cuda)
        d.load_state_dict(first.state_dict())
        decs.append(d)
    return decs


@pytest.mark.parametrize("cd", [None, torch.bfloat16], ids=["fp32", "bf16"])
def test_persistent_decodes_on_several_streams_equal_the_serial_ones(cuda, cd):
    """Round-5 review: two (three) persistent decodes in flight on different streams could each get part of their clusters
    resident and starve.  tpspp_nrtr_decoder_fwd now orders every persistent decode of a device behind the previous one
    (an event, no host synchronisation): decodes issued back to back on three streams -- from one thread and from three
    threads -- give exactly the serial results and status 0; nothing is NaN."""
    import threading
    n = 512
    decs = _decoders(cuda, 3)
    for d in decs:
        d.compute_dtype = cd
    encs = [torch.randn(n, 64, 512, device=cuda) for _ in decs]
    with torch.no_grad():
        serial = [d(None, e, None, None, train_mode=False).clone() for d, e in zip(decs, encs)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=cuda) for _ in decs]
        for rep in range(2):
            outs = [None] * len(decs)

            def work(i):
                with torch.no_grad(), torch.cuda.stream(streams[i]):
                    outs[i] = decs[i](None, encs[i], None, None, train_mode=False)

            if rep == 0:
                for i in range(len(decs)):
                    work(i)
            else:
                ts = [threading.Thread(target=work, args=(i,)) for i in range(len(decs))]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
            torch.cuda.synchronize()
            for i, d in enumerate(decs):
                d.check_status()
                assert torch.isfinite(outs[i]).all(), (rep, i)
                assert torch.equal(outs[i], serial[i]), (rep, i, float((outs[i] - serial[i]).abs().max()))


def test_persistent_decode_on_a_partly_occupied_device(cuda):
    """A decode while another stream holds 96 CUs for 60 ms (tpspp_lab_occupy: one workgroup per CU with 100 KB of LDS, so no
    decoder workgroup fits beside it): the decode's clusters wait for their CUs -- the barrier timeout is wall-clock time,
    4 s by default -- and the result equals the unobstructed one; status 0, nothing NaN."""
    from tps_pp_amd import _lib
    n = 512
    (dec,) = _decoders(cuda, 1)
    enc = torch.randn(n, 64, 512, device=cuda)
    with torch.no_grad():
        want = dec(None, enc, None, None, train_mode=False).clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=cuda)
        for wgs in (96, 200):
            _lib.check(_lib.lib().tpspp_lab_occupy(wgs, 100 * 1024, 60, side.cuda_stream), "tpspp_lab_occupy")
            got = dec(None, enc, None, None, train_mode=False)
            torch.cuda.synchronize()
            dec.check_status()
            assert torch.isfinite(got).all() and torch.equal(got, want), wgs


def test_persistent_decode_timeout_is_loud(cuda):
    """The failure path on demand (TPSPP_HEAD_TEST_STALL: one workgroup of the first cluster sits out two timeouts at step 2,
    TPSPP_HEAD_TIMEOUT_MS=20): the status word is 1, `AttnConvertor.tensor2idx` / `NRTRDecoder.check_status` raise
    TpsppError instead of decoding NaN into strings, the stalled cluster's scores are NaN from step 2 to the LAST step (no
    step is left uninitialised), every other cluster's are untouched -- and the next decode is clean."""
    import os
    from tps_pp_amd import _lib
    n, seq = 100, 9
    (dec,) = _decoders(cuda, 1, seq)
    conv = AttnConvertor(dict_type="DICT90", with_unknown=True, max_seq_len=seq)
    enc = torch.randn(n, 64, 512, device=cuda)
    with torch.no_grad():
        want = dec(None, enc, None, None, train_mode=False).clone()
        want_idx = conv.tensor2idx(want)
        os.environ["TPSPP_HEAD_TEST_STALL"] = "2"
        os.environ["TPSPP_HEAD_TIMEOUT_MS"] = "20"
        try:
            out = dec(None, enc, None, None, train_mode=False)
        finally:
            del os.environ["TPSPP_HEAD_TEST_STALL"], os.environ["TPSPP_HEAD_TIMEOUT_MS"]
        with pytest.raises(_lib.TpsppError, match="timed out"):
            conv.tensor2idx(out)
        with pytest.raises(_lib.TpsppError, match="timed out"):
            dec.check_status()
        assert int(dec.last_status.cpu()[0]) == 1
        o = out.cpu()
        assert torch.equal(o[:32, :2], want[:32, :2].cpu())                  # steps before the stall completed
        assert torch.isnan(o[:32, 2:]).all()                                 # the stalled cluster: NaN from step 2 to the end
        assert torch.equal(o[32:], want[32:].cpu())                          # the other clusters never noticed
        again = dec(None, enc, None, None, train_mode=False)
        assert conv.tensor2idx(again) == want_idx and torch.equal(again, want)


def test_persistent_decode_respects_the_devices_capacity(cuda):
    """Co-residency guard: a device that cannot hold a group of 8 clusters (TPSPP_HEAD_PERSIST_GROUPS=0 stands in for a small
    or CU-masked part) gets the launch pipeline -- for the exact-fp32 head the same bits as TPSPP_HEAD_NO_PERSIST=1 --, one that
    holds a single group decodes 256 images per launch: same scores as the default in both cases."""
    import os
    n, seq = 557, 5
    (dec,) = _decoders(cuda, 1, seq)
    enc = torch.randn(n, 64, 512, device=cuda)
    metas = [dict(valid_ratio=(1.0, 0.5)[i % 2]) for i in range(n)]
    with torch.no_grad():
        want = dec(None, enc, None, metas, train_mode=False).clone()
        try:
            os.environ["TPSPP_HEAD_NO_PERSIST"] = "1"
            pipeline = dec(None, enc, None, metas, train_mode=False).clone()
            del os.environ["TPSPP_HEAD_NO_PERSIST"]
            for groups in ("0", "1"):
                os.environ["TPSPP_HEAD_PERSIST_GROUPS"] = groups
                got = dec(None, enc, None, metas, train_mode=False)
                dec.check_status()
                assert torch.equal(got, pipeline if groups == "0" else want), groups
        finally:
            os.environ.pop("TPSPP_HEAD_NO_PERSIST", None)
            os.environ.pop("TPSPP_HEAD_PERSIST_GROUPS", None)
    assert torch.equal(pipeline, want)                                        # (exact-fp32 head: bit-identical routes)


@pytest.mark.parametrize("n,L,C", [(5, 40, 92), (67, 70, 92), (3, 1, 7), (33, 64, 130)])
def test_attn_tensor2idx_kernel_against_the_reference_scan(cuda, n, L, C):
    """`tpspp_attn_tensor2idx_fwd` (arg-max with torch.max's first-index tie rule, NaN beating numbers, skip <PAD>, stop at
    the first <EOS>) against the reference's own composition on the CPU (convertors/attn.py:124-140 = the CPU branch of
    `AttnConvertor.tensor2idx`): identical index lists and bit-identical scores; ties, rows of -inf, NaN, more than 64
    positions."""
    g = torch.Generator().manual_seed(n * 1000 + L)
    conv = AttnConvertor(dict_type="DICT90", with_unknown=True, max_seq_len=L)
    conv.end_idx, conv.padding_idx = min(conv.end_idx, C - 2), min(conv.padding_idx, C - 1)
    x = torch.rand((n, L, C), generator=g)
    x = (x * 8).round() / 8                                                    # many exact ties
    x[:, :, conv.end_idx] += (torch.rand((n, L), generator=g) < 0.08).float()      # an <EOS> here and there
    x[:, :, conv.padding_idx] += (torch.rand((n, L), generator=g) < 0.1).float()   # <PAD> before it
    if n > 2 and L > 1:
        x[1, 0] = float("-inf")
        x[2, L - 1, 3] = float("nan")
        x[2, L - 1, 5] = float("nan")
    want_i, want_s = conv.tensor2idx(x)
    got_i, got_s = conv.tensor2idx(x.to(cuda))
    assert got_i == want_i
    assert len(got_s) == len(want_s)
    for a, b in zip(got_s, want_s):
        assert np.array_equal(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), equal_nan=True)


def test_eval_mode_with_autograd_enabled_says_so_once(cuda, caplog):
    """Frozen-BatchNorm fine-tuning (model.eval(), grad enabled, trainable parameters) runs the HIP inference kernels, whose
    outputs carry no graph: every module that takes that path says so once instead of silently returning detached tensors
    (round-5 ADVICE; TPS_PP already did)."""
    import logging
    from tps_pp_amd import ResNetABI_v2_large
    enc, dec = small_modules(cuda)
    feat = torch.zeros(2, cases.HD_SMALL["d_model"], 2, 4, device=cuda)
    with caplog.at_level(logging.WARNING, logger="tps_pp_amd"):
        for _ in range(2):
            out_enc = enc(feat, None)
            out = dec(feat, out_enc, None, None, train_mode=False)
    msgs = [r.getMessage() for r in caplog.records if "records no graph" in r.getMessage()]
    assert sum("NRTREncoder" in m for m in msgs) == 1 and sum("NRTRDecoder" in m for m in msgs) == 1
    assert not out.requires_grad
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="tps_pp_amd"), torch.no_grad():
        enc2, _ = small_modules(cuda)
        enc2(feat, None)
    assert not [r for r in caplog.records if "records no graph" in r.getMessage()]      # no_grad inference: silent
