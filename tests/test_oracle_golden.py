"""CPU tests: the oracle against the reference's own outputs (tests/golden/*.npz).

The grid generator and sampler are C with explicit fmaf(): bit-exact on any host.  The regressor /
backbone oracles are PyTorch-CPU convolutions whose blocking may differ between CPU models, so they
are held to a tight tolerance here (they were bit-exact in the container that generated the
fixtures: make_golden.py asserts it)."""
import numpy as np
import pytest
import torch

import cases
from oracle import tpspp_oracle as TO

CONV_TOL = 2e-5
EXACT_HOST = cases.on_generating_host()


def close(a, b, tol=CONV_TOL, pinned=True):
    """PyTorch-CPU oracles: bit-identical to the fixtures in the container that generated them (make_golden.py
    asserts it), to rounding on other hosts (MKL / oneDNN pick other kernels per CPU model and thread count)."""
    a, b = np.asarray(a), np.asarray(b)
    if EXACT_HOST and pinned:
        return np.array_equal(a, b)
    return a.shape == b.shape and float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max()))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def biteq(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def tsd(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def test_constants_match_reference(oracle):
    K = cases.load("constants")
    c = oracle.classic_constants(cases.CL_F, cases.CL_HW)
    p = oracle.tpspp_constants(cases.PP_HW, cases.PP_POINT)
    assert np.array_equal(c["C"], K["classic_C"]) and np.array_equal(c["P"], K["classic_P"])
    assert np.array_equal(p["C"], K["pp_C"]) and np.array_equal(p["P"], K["pp_P"])
    # np.linalg.inv / np.log may differ in the last ulp between LAPACK / libm builds
    np.testing.assert_allclose(c["inv_delta_C"], K["classic_inv_delta_C"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c["P_hat"], K["classic_P_hat"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(p["hat_C"], K["pp_hat_C"], rtol=1e-5, atol=2e-4)   # entries up to 223
    np.testing.assert_allclose(p["P_hat"], K["pp_P_hat"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(oracle.classic_initial_ctrl(20), cases.classic_initial_ctrl(20))
    assert np.array_equal(oracle.tpspp_initial_ctrl((2, 16)), cases.tpspp_initial_ctrl((2, 16)))


def test_classic_grid_and_sampler_bit_exact(oracle):
    K, G, inp = cases.load("constants"), cases.load("classic_warp"), cases.g2_inputs()
    T = oracle.solve_T(K["classic_inv_delta_C"], inp["ctrl"])
    grid = oracle.build_grid(K["classic_P_hat"], T)
    assert biteq(grid, G["grid"])
    out, idx = oracle.grid_sample(inp["img"], grid, cases.CL_HW, return_idx=True)
    assert biteq(out, G["out"])
    assert biteq(oracle.grid_sample(inp["img_smooth"], grid, cases.CL_HW), G["out_smooth"])
    # indices are consistent with the grid: floor of the clamped un-normalised coordinate
    H, W = cases.CL_HW
    ix = np.clip(((grid[..., 0] + np.float32(1)) * np.float32(0.5)) * np.float32(W - 1), 0, W - 1)
    assert np.array_equal(idx[..., 0], np.floor(ix).astype(np.int32))
    # the other two weight forms stay within float rounding of the reference
    for wf in (0, 1):
        assert np.abs(oracle.grid_sample(inp["img"], grid, cases.CL_HW, wf) - G["out"]).max() < 5e-7
    # fused entry == pieces
    r = oracle.warp(inp["img"], inp["ctrl"], K["classic_inv_delta_C"], K["classic_P_hat"],
                    cases.CL_HW, want_grid=True, want_idx=True)
    assert biteq(r["grid"], G["grid"]) and biteq(r["out0"], G["out"]) and np.array_equal(r["idx"], idx)
    # the border clamp was exercised by the large perturbations
    assert (np.abs(G["grid"]) > 1.0).mean() > 0.02


def test_tpspp_warp_stage_bit_exact(oracle):
    K, G, inp = cases.load("constants"), cases.load("tpspp_warp"), cases.g3_inputs()
    r = oracle.warp(inp["feat_grid"], inp["ctrl"], K["pp_hat_C"], K["pp_P_hat"], cases.PP_HW,
                    P_xy=K["pp_P"].astype(np.float32), score=inp["score"], in1=inp["x"],
                    want_grid=True)
    assert biteq(r["grid"], G["grid"])
    assert biteq(r["out0"], G["output"]) and biteq(r["out1"], G["mp_img"])


def _classic_sd():
    import json, os
    keys = json.load(open(os.path.join(cases.HERE, "state_dict_keys.json")))
    shapes = keys["TPSPreprocessor(20,(32,100),(32,100),3)"]
    sd = cases.synth_state({k: np.empty(v) for k, v in shapes.items()}, 1, cases.g1_state_rule,
                           cases.G1_KEEP)
    K = cases.load("constants")
    sd["LocalizationNetwork.localization_fc2.bias"] = cases.classic_initial_ctrl(20).reshape(-1)
    sd["GridGenerator.inv_delta_C"] = K["classic_inv_delta_C"]
    sd["GridGenerator.P_hat"] = K["classic_P_hat"]
    return sd


def test_classic_module_oracle(oracle):
    G = cases.load("classic_module")
    sd = _classic_sd()
    o = TO.classic_forward(tsd(sd), cases.g1_inputs()["img"])
    np.testing.assert_allclose(o["ctrl"], G["ctrl"], atol=CONV_TOL, rtol=0)
    # the transformation stage from the reference's own control points: bit-exact
    r = oracle.warp(cases.g1_inputs()["img"], G["ctrl"], sd["GridGenerator.inv_delta_C"],
                    sd["GridGenerator.P_hat"], cases.CL_HW, want_grid=True)
    assert biteq(r["grid"], G["grid"]) and biteq(r["out0"], G["out"])


def _tpspp_sd(variant):
    import json, os
    shapes = json.load(open(os.path.join(cases.HERE, "state_dict_keys.json")))["TPS_PP"]
    if variant == "ResNet45":
        shapes = {k: v for k, v in shapes.items()
                  if not k.startswith(("down0_1.", "down1_1.", "down_feat."))}
        shapes["down0.conv.weight"] = [64, 32, 3, 3]
    sd = cases.synth_state({k: np.empty(v) for k, v in shapes.items()}, 4, cases.tpspp_state_rule,
                           cases.TPSPP_KEEP)
    K = cases.load("constants")
    sd["TPE.localization_fc2.bias"] = cases.tpspp_initial_ctrl().reshape(-1)
    sd["atten_tps.hat_C"], sd["atten_tps.P_hat"] = K["pp_hat_C"], K["pp_P_hat"]
    return sd


@pytest.mark.parametrize("variant,fname", [("ResNet45v2", "tpspp_module_v2"),
                                           ("ResNet45", "tpspp_module_v1")])
def test_tpspp_module_oracle(oracle, variant, fname):
    G = cases.load(fname)
    sd = _tpspp_sd(variant)
    inp = cases.g4_inputs(variant)
    with torch.no_grad():
        ctrl, score, feat_grid, inter = TO.tpspp_regress(tsd(sd), inp["x"], inp["outs"], variant)
    np.testing.assert_allclose(ctrl.numpy(), G["ctrl"], atol=CONV_TOL, rtol=0)
    np.testing.assert_allclose(score.numpy(), G["pc_score"], atol=CONV_TOL, rtol=0)
    if variant == "ResNet45v2":
        np.testing.assert_allclose(cases.sub(inter["feat_cat"].numpy()), G["feat_cat_sub"], atol=CONV_TOL)
        np.testing.assert_allclose(inter["cbam"].numpy(), G["cbam"], atol=CONV_TOL)
        np.testing.assert_allclose(cases.sub(inter["dgab"].numpy()), G["dgab_sub"], atol=1e-4)
        for i in range(4):
            np.testing.assert_allclose(cases.sub(inter[f"enc{i}"].numpy()), G[f"enc{i}_sub"], atol=CONV_TOL)
    # transformation stage from the reference's control points and score
    K = cases.load("constants")
    P_xy = K["pp_P"].astype(np.float32)
    r = oracle.warp(feat_grid.numpy(), G["ctrl"], sd["atten_tps.hat_C"], sd["atten_tps.P_hat"],
                    cases.PP_HW, P_xy=P_xy, score=G["pc_score"], in1=inp["x"], want_grid=True)
    assert biteq(r["grid"], G["grid"])
    assert biteq(r["out1"], G["mp_img"])                 # samples the raw input: exact on any host
    if variant == "ResNet45":
        assert biteq(r["out0"], G["output"])             # feat_grid is the raw input here
    else:
        assert np.abs(r["out0"] - G["output"]).max() <= 1e-4


def test_backbone_stem_oracle():
    G = cases.load("backbone_stem")
    import json  # shapes come from the golden arrays + the architecture (resnet_v2_large.py:77-135)
    shapes = {"conv1.weight": (32, 3, 3, 3), "conv1.bias": (32,)}
    for p in ("bn1",):
        shapes.update({f"{p}.weight": (32,), f"{p}.bias": (32,), f"{p}.running_mean": (32,),
                       f"{p}.running_var": (32,)})

    def block(prefix, cin, cout, down):
        shapes[f"{prefix}.conv1.weight"] = (cout, cin, 1, 1)
        shapes[f"{prefix}.conv2.weight"] = (cout, cout, 3, 3)
        for b in ("bn1", "bn2"):
            for s_ in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"{prefix}.{b}.{s_}"] = (cout,)
        if down:
            shapes[f"{prefix}.downsample.0.weight"] = (cout, cin, 1, 1)
            for s_ in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"{prefix}.downsample.1.{s_}"] = (cout,)
    # layer1: 3 blocks 32->32 stride 1 (no downsample); layer2: 4 blocks 32->64 stride 2
    for b in range(3):
        block(f"layer1.{b}", 32, 32, False)
    for b in range(4):
        block(f"layer2.{b}", 32 if b == 0 else 64, 64, b == 0)
    sd = cases.synth_state({k: np.empty(v) for k, v in shapes.items()}, 7, cases.backbone_state_rule)
    x, outs = TO.backbone_stem(tsd(sd), cases.g7_inputs()["img"])
    np.testing.assert_allclose(x.numpy(), G["x"], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(outs[0].numpy()[:, ::4], G["outs0_sub"], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(outs[1].numpy()[:, ::4], G["outs1_sub"], atol=1e-4, rtol=1e-5)


def test_oracle_edge_cases(oracle):
    """Shapes the reference's own test uses (test_ocr_preprocessor.py:19-29: 1x1x32x100) and
    degenerate ones: every grid point outside [-1,1] must give a border pixel, exactly."""
    Kc = oracle.classic_constants(20, (32, 100))
    img = np.arange(32 * 100, dtype=np.float32).reshape(1, 1, 32, 100)
    far = np.full((1, 20, 2), 50.0, np.float32)          # all control points far to the south-east
    r = oracle.warp(img, far, Kc["inv_delta_C"], Kc["P_hat"], (32, 100), want_idx=True)
    assert r["out0"].shape == (1, 1, 32, 100)
    assert np.all(r["out0"] == img[0, 0, 31, 99])
    assert np.all(r["idx"][..., 0] == 99) and np.all(r["idx"][..., 1] == 31)
    nan = np.full((1, 20, 2), np.nan, np.float32)        # NaN coordinates clamp to pixel (0, 0)
    r = oracle.warp(img, nan, Kc["inv_delta_C"], Kc["P_hat"], (32, 100), want_idx=True)
    assert np.all(r["idx"] == 0)
    empty = oracle.warp(np.zeros((0, 3, 32, 100), np.float32), np.zeros((0, 20, 2), np.float32),
                        Kc["inv_delta_C"], Kc["P_hat"], (32, 100))
    assert empty["out0"].shape == (0, 3, 32, 100)


# ---- recogniser head: oracle/nrtr_oracle.py vs the reference's outputs -----
def _head_state(kind, small):
    """The synthetic state_dict the fixtures were generated with, rebuilt without the reference: key
    names and shapes from the mirror modules (their layout is pinned in tests/test_host_modules.py)."""
    import tps_pp_amd as P
    cfg = dict(cases.HD_SMALL) if small else {}
    if kind == "enc":
        m, seed = P.NRTREncoder(**cfg), 9
    else:
        extra = dict(d_embedding=cfg["d_model"]) if small else {}
        m, seed = P.NRTRDecoder(num_classes=cases.NUM_CLASSES, start_idx=cases.START_IDX,
                                padding_idx=cases.PAD_IDX, max_seq_len=cases.HD_MAXLEN if small else 40,
                                **cfg, **extra), 10
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    for k, v in cases.synth_state(m.state_dict(), seed, cases.head_state_rule, cases.HD_KEEP).items():
        sd[k] = torch.from_numpy(v)
    return sd


def test_nrtr_encoder_oracle():
    from oracle import nrtr_oracle as NO
    G = cases.load("nrtr_encoder")
    sd = _head_state("enc", True)
    feat = cases.g9_inputs()["feat"]
    nh = cases.HD_SMALL["n_head"]
    assert close(NO.encoder_forward(sd, feat, nh, cases.HD_RATIOS).numpy(), G["out_masked"])
    assert close(NO.encoder_forward(sd, feat, nh, None).numpy(), G["out_nomask"])


def test_nrtr_decoder_oracle():
    from oracle import nrtr_oracle as NO
    G = cases.load("nrtr_decoder")
    sd = _head_state("dec", True)
    inp = cases.g10_inputs()
    nh = cases.HD_SMALL["n_head"]
    assert np.array_equal(sd["position_enc.position_table"].numpy(), NO.sinusoid_table(200, 128).numpy())
    lo = NO.decoder_forward_train(sd, inp["out_enc"], inp["padded_targets"], nh, cases.PAD_IDX, cases.HD_RATIOS)
    assert close(lo.numpy(), G["logits"])
    pr = NO.decoder_forward_test(sd, inp["out_enc"], nh, cases.HD_MAXLEN, cases.START_IDX, cases.PAD_IDX,
                                 cases.HD_RATIOS)
    assert close(pr.numpy(), G["probs"]) and np.array_equal(pr.numpy().argmax(-1), G["probs"].argmax(-1))
    pr = NO.decoder_forward_test(sd, inp["out_enc"], nh, cases.HD_MAXLEN, cases.START_IDX, cases.PAD_IDX, None)
    assert close(pr.numpy(), G["probs_nomask"])


def test_nrtr_head_full_oracle_and_convertor():
    from oracle import nrtr_oracle as NO
    G = cases.load("nrtr_head_full")
    o = NO.head_simple_test(_head_state("enc", False), _head_state("dec", False), cases.g11_inputs()["feat"])
    assert close(o["out_enc"].numpy()[:, :, ::8], G["out_enc_sub"])
    assert close(o["out_dec"].numpy(), G["out_dec"])
    assert o["text"] == [str(s) for s in G["text"]]
    assert [len(i) for i in o["indexes"]] == G["idx_len"].tolist()
    idx2char, unk, start, end, pad = NO.attn_dictionary()
    assert (unk, start, end, pad, len(idx2char)) == (90, cases.START_IDX, cases.END_IDX, cases.PAD_IDX,
                                                     cases.NUM_CLASSES)


def test_recognizer_end_to_end_oracle():
    """backbone (+TPS++) -> encoder -> greedy decoder -> convertor, oracle vs the reference's outputs."""
    import tps_pp_amd as P
    from oracle import tpspp_oracle as TO
    G = cases.load("recognizer_e2e")

    def synth_sd(m, seed, rule, keep=()):
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        for k, v in cases.synth_state(m.state_dict(), seed, rule, keep).items():
            sd[k] = torch.from_numpy(v)
        return sd
    bb = synth_sd(P.ResNetABI_v2_large(arch_settings=[3, 4, 6, 6, 3], strides=cases.G12_STRIDES), 7,
                  cases.backbone_state_rule)
    tps = synth_sd(P.TPS_PP(variant="ResNet45"), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    o = TO.recognizer_simple_test(bb, tps, _head_state("enc", False), _head_state("dec", False),
                                  cases.g12_inputs()["img"], cases.G12_WIDTHS)
    assert close(o["feat"].numpy()[:, ::8], G["feat_sub"])
    assert close(o["out_dec"].numpy(), G["out_dec"], 1e-4)
    assert o["text"] == [str(s) for s in G["text"]]
    assert close(np.array(o["scores"][0], dtype=np.float32), G["score0"], 1e-4)


def test_warp_backward_oracle_against_reference(oracle):
    """Row F2: the backward oracle against the reference's own autograd results (golden G14): both variants
    of the oracle (grid through torch.bmm -- bit-identical to the fixtures where they were generated, host
    dependent elsewhere -- and grid through the FMA chain with the two products transposed in float64).
    The parameter gradients are sums of 1024-3200 terms amplified by inv_delta_C (+-223), so an fp32
    accumulation differs from a float64 one by ~1e-4 of the largest entry."""
    G = cases.load("warp_backward")
    gi = cases.g14_inputs()
    inp = cases.g2_inputs()
    c = oracle.classic_constants(cases.CL_F, cases.CL_HW)
    for chain in (False, True):
        o = oracle.warp_backward(gi["g_out_cl"], inp["img_smooth"], inp["ctrl"], c["inv_delta_C"], c["P_hat"],
                                 cases.CL_HW, chain_grid=chain)
        # (G2 holds lattices perturbed by up to +-2: there the grid out of torch.bmm itself moves by 1e-4
        # between hosts, and so does every gradient that is scattered through it)
        assert close(o["g_in0"], G["cl_g_img"], 1e-2 if not chain else CONV_TOL, not chain), chain
        assert close(o["g_ctrl"], G["cl_g_ctrl"], 1e-2 if not chain else 5e-4, not chain), chain
    inp = cases.g3_inputs()
    c = oracle.tpspp_constants(cases.PP_HW, cases.PP_POINT)
    kw = dict(P_xy=c["P_xy"], score=inp["score"], in1=inp["x"], g_out1=gi["g_out1"])
    for chain in (False, True):
        o = oracle.warp_backward(gi["g_out0"], inp["feat_grid"], inp["ctrl"], c["hat_C"], c["P_hat"], cases.PP_HW,
                                 chain_grid=chain, **kw)
        t_in, t_par = (CONV_TOL, 5e-4) if chain else (1e-3, 1e-2)      # the bmm variant is host dependent
        assert close(cases.sub(o["g_in0"]), G["pp_g_feat_grid_sub"], t_in, not chain), chain
        assert close(cases.sub(o["g_in1"]), G["pp_g_x_sub"], t_in, not chain), chain
        assert close(o["g_ctrl"], G["pp_g_ctrl"], t_par, not chain), chain
        assert close(o["g_score"], G["pp_g_score"], t_par, not chain), chain


def test_c_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The C oracle (test infrastructure: the checker of every bit-exact parity claim) built with
    -fsanitize=address,undefined and run in a child process (libasan preloaded in front of the Python interpreter) over the
    shapes the parity tests use it on and the edge cases the sampler has: classic and TPS_PP geometry (two inputs, score),
    C = 1, a one-pixel-high input, grids far outside [-1, 1], NaN / inf control points, optional outputs on and off.  No GPU,
    no sanitizer on the GPU side (this pool refuses those): a heap overflow or an out-of-range index in the checker would
    otherwise surface as a parity failure of a correct kernel -- or hide a real one."""
    import os
    import shutil
    import subprocess
    import sys
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libtps_oracle_asan.so")
    subprocess.check_call([gcc, "-O1", "-g", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2",
                           "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared",
                           "-o", so, os.path.join(root, "oracle", "tps_oracle.c"), "-lm"])
    code = """
import numpy as np
from oracle import tps_oracle as O
rng = np.random.default_rng(0)
for hw, C, N in (((32, 100), 3, 3), ((32, 100), 1, 1), ((16, 36), 4, 2), ((64, 200), 3, 1)):
    K = O.classic_constants(20, hw)
    img = rng.standard_normal((N, C) + hw).astype(np.float32)
    for amp in (0.05, 0.6, 5.0):
        ctrl = (O.classic_initial_ctrl(20)[None] + amp * rng.standard_normal((N, 20, 2))).astype(np.float32)
        r = O.warp(img, ctrl, K["inv_delta_C"], K["P_hat"], hw, want_grid=True, want_idx=True)
        assert r["out0"].shape == (N, C) + hw
        O.warp(img, ctrl, K["inv_delta_C"], K["P_hat"], hw)
    bad = O.classic_initial_ctrl(20)[None].repeat(N, 0).astype(np.float32)
    bad[0, 3, 0] = np.nan; bad[0, 5, 1] = np.inf
    O.warp(img, bad, K["inv_delta_C"], K["P_hat"], hw, want_grid=True, want_idx=True)
Kp = O.tpspp_constants((16, 64), (2, 16))
fg = rng.standard_normal((2, 5, 32, 128)).astype(np.float32)
x = rng.standard_normal((2, 3, 16, 64)).astype(np.float32)
thin = rng.standard_normal((2, 2, 1, 64)).astype(np.float32)          # a one-pixel-high input: no south row anywhere
ctrl = (O.tpspp_initial_ctrl((2, 16))[None] + 0.3 * rng.standard_normal((2, 32, 2))).astype(np.float32)
score = np.tanh(rng.standard_normal((2, 1024, 32))).astype(np.float32)
O.warp(fg, ctrl, Kp["hat_C"], Kp["P_hat"], (16, 64), P_xy=Kp["P_xy"], score=score, in1=x, want_grid=True, want_idx=True)
O.warp(thin, ctrl, Kp["hat_C"], Kp["P_hat"], (16, 64), P_xy=Kp["P_xy"], score=score, in1=x)
T = O.solve_T(Kp["hat_C"], ctrl)
g = O.build_grid(Kp["P_hat"], T, Kp["P_xy"], score)
O.grid_sample(fg, (50.0 * g).reshape(2, 16, 64, 2), (16, 64), return_idx=True)
O.grid_sample(thin, g, (16, 64), weight_form=1)
print("sanitized oracle ok")
"""
    env = dict(os.environ, LD_PRELOAD=asan, TPS_ORACLE_SO=so, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1", PYTHONPATH=root, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
