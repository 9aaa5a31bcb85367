"""-m gpu: the remaining BASELINE.json configurations at their full per-GPU sizes, through
size-independent properties (the oracle cannot run these sizes in seconds):

* configs[2]  regressor + warp (`TPS_PP.forward`) at batch 1024, fp32 and bf16 (bf16 tensors at the module boundary,
  bf16 MFMA convolutions; checked against the bf16-emulating CPU oracle and, loosely, the fp32 one);
* configs[3]  backbone (stem, layer1-2, TPS++, layer3-5) at 256 images per GPU (2048 over 8 GPUs);
* configs[4]  image -> string at 256 images per GPU.

Property: every image is processed independently, so the rows of a full-size batch equal the same
images run as a small batch (bit for bit where both batches take the same kernels, to fp32 rounding
where the launch heuristics pick a different tiling); the small batch in turn is checked against the
CPU oracle (values within 1e-4, strings identical)."""
import numpy as np
import pytest
import torch

import cases
from oracle import tpspp_oracle as TO
from tps_pp_amd import synth
from test_gpu_head import build_recognizer

pytestmark = pytest.mark.gpu
TOL = 1e-4
PICK = [0, 1, 129, 255]


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def tile_batch(a, n):
    """n images from a small seeded set, each with its own gain so that rows differ."""
    reps = (n + a.shape[0] - 1) // a.shape[0]
    out = np.concatenate([a * (1.0 - 0.003 * r) for r in range(reps)], 0)[:n]
    return np.ascontiguousarray(out.astype(np.float32))


def test_config2_regressor_and_warp_batch_1024(cuda):
    from tps_pp_amd import TPS_PP
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    cpu_sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(cuda)
    n = 1024
    inp = cases.g4_inputs("ResNet45v2")
    x = tile_batch(inp["x"], n)
    outs = [tile_batch(o, n) for o in inp["outs"]]
    pick = [0, 1, 513, 1023]
    with torch.no_grad():
        full = m(dev(x, cuda), [dev(o, cuda) for o in outs])
        small = m(dev(x[pick], cuda), [dev(o[pick], cuda) for o in outs])
    for k in ("output", "mp_img", "pc_score"):
        assert torch.equal(full[k][pick], small[k]), k
    o = TO.tpspp_forward(cpu_sd, x[pick[:2]], [o_[pick[:2]] for o_ in outs], "ResNet45v2")
    assert np.abs(small["output"][:2].cpu().numpy() - o["output"]).max() <= TOL
    assert np.abs(small["mp_img"][:2].cpu().numpy() - o["mp_img"]).max() <= TOL


def test_config2_bf16x3_regressor_and_warp_batch_1024(cuda):
    """configs[2] geometry at batch 1024 with `compute_dtype = "bf16x3"` (fp32 tensors, three-term bf16 split): rows of
    the full batch equal the small batch bit for bit, the small batch meets the exact path's 1e-4 against the fp32 CPU
    oracle, and the full batch stays within 1e-4 of the exact-fp32 HIP kernels on every image."""
    from tps_pp_amd import TPS_PP
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    cpu_sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(cuda)
    n = 1024
    inp = cases.g4_inputs("ResNet45v2")
    x = tile_batch(inp["x"], n)
    outs = [tile_batch(o, n) for o in inp["outs"]]
    pick = [0, 1, 513, 1023]
    with torch.no_grad():
        exact = m(dev(x, cuda), [dev(o, cuda) for o in outs])
        m.compute_dtype = "bf16x3"
        full = m(dev(x, cuda), [dev(o, cuda) for o in outs])
        small = m(dev(x[pick], cuda), [dev(o[pick], cuda) for o in outs])
    for k in ("output", "mp_img", "pc_score"):
        assert torch.equal(full[k][pick], small[k]), k
        assert (full[k] - exact[k]).abs().max().item() <= TOL, k
    o = TO.tpspp_forward(cpu_sd, x[pick[:2]], [o_[pick[:2]] for o_ in outs], "ResNet45v2")
    assert np.abs(small["output"][:2].cpu().numpy() - o["output"]).max() <= TOL
    assert np.abs(small["mp_img"][:2].cpu().numpy() - o["mp_img"]).max() <= TOL


def test_config2_bf16_regressor_and_warp_batch_1024(cuda):
    """BASELINE.json configs[2] as named: batch 1024, bf16.  Rows of the full batch equal the same images run
    as a small batch bit for bit; the small batch is checked against the CPU oracle that rounds to bfloat16
    at the same places (outputs within one bf16 ulp of the oracle's rounded value, control points to 1e-5)
    and against the fp32 oracle at bf16 resolution."""
    from tps_pp_amd import TPS_PP
    m = TPS_PP().eval()
    sd = cases.synth_state(m.state_dict(), 4, cases.tpspp_state_rule, cases.TPSPP_KEEP)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    cpu_sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(cuda)
    n = 1024
    inp = cases.g4_inputs("ResNet45v2")
    b16 = lambda a: torch.from_numpy(a).to(torch.bfloat16)      # noqa: E731
    x = b16(tile_batch(inp["x"], n))
    outs = [b16(tile_batch(o, n)) for o in inp["outs"]]
    pick = [0, 1, 513, 1023]
    with torch.no_grad():
        full = m(x.to(cuda), [o.to(cuda) for o in outs])
        small = m(x[pick].to(cuda), [o[pick].to(cuda) for o in outs])
        ctrl, score, _ = m.regress(x[pick[:2]].to(cuda), [o[pick[:2]].to(cuda) for o in outs])
    assert full["output"].dtype == torch.bfloat16 and full["mp_img"].dtype == torch.bfloat16
    for k in ("output", "mp_img", "pc_score"):
        assert torch.equal(full[k][pick], small[k]), k
    xs, os_ = x[pick[:2]].float().numpy(), [o[pick[:2]].float().numpy() for o in outs]
    ob = TO.tpspp_forward(cpu_sd, xs, os_, "ResNet45v2", bf16=True)
    of = TO.tpspp_forward(cpu_sd, xs, os_, "ResNet45v2", bf16=False)
    assert np.abs(ctrl.cpu().numpy() - ob["ctrl"]).max() <= 1e-5
    assert np.abs(score.float().cpu().numpy() - ob["pc_score"]).max() <= 5e-4
    assert np.abs(ctrl.cpu().numpy() - of["ctrl"]).max() <= 1e-4
    for k in ("output", "mp_img"):
        got = small[k][:2].float().cpu().numpy()
        for o, slack in ((ob, 1e-4), (of, 1e-2)):
            ref = o[k]
            assert (np.abs(got - ref) <= np.abs(ref) * 2.0 ** -8 + slack * np.abs(ref).max()).all(), k


def test_config3_and_4_backbone_and_strings_256_per_gpu(cuda):
    m = build_recognizer(cuda)
    n = 256
    img = tile_batch(synth.smooth_image((8, 3, 32, 128), "cfg34.img", 3), n)
    widths = [128 if i % 3 else 96 for i in range(n)]
    metas = [dict(resize_shape=(32, w, 3)) for w in widths]
    with torch.no_grad():
        feat_full = m.extract_feat(dev(img, cuda), test=True)["output"]
        res_full = m(dev(img, cuda), [dict(mm) for mm in metas], return_loss=False)
        feat_small = m.extract_feat(dev(img[PICK], cuda), test=True)["output"]
        res_small = m(dev(img[PICK], cuda), [dict(metas[i]) for i in PICK], return_loss=False)
    assert feat_full.shape == (n, 512, 4, 16)
    # configs[3]: batch independence (to rounding: the 1x1 projections of a 4-image batch take the split-K
    # kernel, those of the 256-image batch the tiled one -- different summation order)
    assert (feat_full[PICK] - feat_small).abs().max() <= 1e-5
    assert [res_full[i]["text"] for i in PICK] == [r["text"] for r in res_small]
    for i, r in zip(PICK, res_small):
        assert np.allclose(res_full[i]["score"], r["score"], rtol=0, atol=1e-5)
    # the small batch against the CPU oracle (two images: the oracle re-runs the decoder 40 times)
    sds = [{k: v.cpu() for k, v in mod.state_dict().items()} for mod in (m.backbone, m.tpsnet, m.encoder, m.decoder)]
    o = TO.recognizer_simple_test(sds[0], sds[1], sds[2], sds[3], img[PICK[:2]], [widths[i] for i in PICK[:2]])
    assert np.abs(feat_small[:2].cpu().numpy() - o["feat"].numpy()).max() <= TOL
    assert [r["text"] for r in res_small[:2]] == o["text"]


def test_config3_and_4_bf16_256_per_gpu(cuda):
    """BASELINE.json configs[3] / [4] in the bf16 configuration at 256 images per GPU: backbone + TPS++ convolutions and
    the head's wide projections on the bf16 matrix cores.  Rows of the full batch equal the same images run as a small
    batch (feature map bit for bit: every layer takes the same kernel at both sizes; strings identical); the feature
    map stays within bf16 resolution of the fp32 CPU oracle's and the strings of the checked images are the oracle's."""
    m = build_recognizer(cuda)
    m.backbone.compute_dtype = m.encoder.compute_dtype = m.decoder.compute_dtype = torch.bfloat16
    n = 256
    img = tile_batch(synth.smooth_image((8, 3, 32, 128), "cfg34.img", 3), n)
    widths = [128 if i % 3 else 96 for i in range(n)]
    metas = [dict(resize_shape=(32, w, 3)) for w in widths]
    with torch.no_grad():
        feat_full = m.extract_feat(dev(img, cuda), test=True)["output"]
        res_full = m(dev(img, cuda), [dict(mm) for mm in metas], return_loss=False)
        feat_small = m.extract_feat(dev(img[PICK], cuda), test=True)["output"]
        res_small = m(dev(img[PICK], cuda), [dict(metas[i]) for i in PICK], return_loss=False)
    assert feat_full.shape == (n, 512, 4, 16) and feat_full.dtype == torch.float32
    assert (feat_full[PICK] - feat_small).abs().max() <= 2e-3 * feat_small.abs().max()
    assert [res_full[i]["text"] for i in PICK] == [r["text"] for r in res_small]
    sds = [{k: v.cpu() for k, v in mod.state_dict().items()} for mod in (m.backbone, m.tpsnet, m.encoder, m.decoder)]
    o = TO.recognizer_simple_test(sds[0], sds[1], sds[2], sds[3], img[PICK[:2]], [widths[i] for i in PICK[:2]])
    ref = o["feat"].numpy()
    err = np.abs(feat_small[:2].cpu().numpy() - ref)
    assert err.max() <= 0.05 * np.abs(ref).max() and err.mean() <= 0.005 * np.abs(ref).max()
    assert [r["text"] for r in res_small[:2]] == o["text"]


@pytest.mark.parametrize("mode,tf_min,word_min,char_min", [(torch.bfloat16, 0.99, 0.72, 0.92), ("bf16x3", 0.9995, 0.97, 0.99)])
def test_config4_reduced_precision_agreement_on_256_distinct_images(cuda, mode, tf_min, word_min, char_min):
    """BASELINE.json configs[4], "word-accuracy parity check", on 256 DISTINCT synthetic crops with random-init weights
    (classifier spread x8 as in bench.py): decisions of the bf16 / bf16x3 configurations against the exact-fp32 kernels
    of the same model -- which other tests hold to 1e-4 / identical strings against the reference and the CPU oracle.
    Thresholds = what was measured minus a margin: bf16 0.995 / 0.80-0.82 / 0.944-0.95 (rounds 2 and 3, driver line
    BENCH_r02.json) -> 0.99 / 0.72 / 0.92; bf16x3 1.0 / 1.0 / 1.0 -> 0.9995 / 0.97 / 0.99.  Random-init decoding is the
    worst case for word agreement (40 characters of noise, one flipped near-tie rewrites the tail)."""
    from tps_pp_amd import metrics
    m = build_recognizer(cuda)
    with torch.no_grad():
        m.decoder.classifier.weight.mul_(8.0)
    n = 256
    img = dev(synth.smooth_image((n, 3, 32, 128), "cfg4.agree", 5), cuda)
    metas = [dict(resize_shape=(32, 128 if i % 3 else 96, 3)) for i in range(n)]
    r = metrics.precision_agreement(m, img, metas, mode)
    print("precision_agreement", mode, r)
    assert r["images"] == n and r["positions"] >= n
    assert r["teacher_forced_self_check_fp32"] >= 0.9999          # forced decoding == greedy decoding in fp32
    assert r["teacher_forced_argmax_agreement"] >= tf_min, r
    assert r["greedy_word_agreement"] >= word_min, r
    assert r["greedy_char_agreement"] >= char_min, r
    # restricted to decisions the fp32 run is sure of (top-2 margin >= 0.05) the arithmetic must agree almost always: this is
    # the figure that means something without a trained checkpoint.  bf16x3 IS the parity configuration of configs[4]
    # (1e-4 on scores, identical strings); plain bf16 is the throughput configuration.
    assert r["positions_margin_ge_0.05"] > 0.5 * r["positions"], r
    assert r["teacher_forced_agreement_margin_ge_0.05"] >= (0.9999 if mode == "bf16x3" else 0.99), r


def test_sharded_recogniser_through_rccl_world_size_1(cuda):
    """configs[3] / [4] path of `bench.py --gpus N` (extra.recognizer_sharded) and tests/tools/eval_e2e.py on the one GPU of
    this box: a real RCCL ("nccl") process group of size 1, the decoder scores gathered with all_gather_into_tensor on
    device tensors, strings converted from the gathered tensor -- equal to the un-sharded recogniser's.  (World size 2 is
    covered on CPU with gloo: tests/test_dist_gloo.py, tests/test_bench_launch.py.)"""
    import os
    import socket
    import torch.distributed as dist
    from tps_pp_amd import dist as tdist
    m = build_recognizer(cuda)
    n = 8
    img = dev(synth.smooth_image((n, 3, 32, 128), "cfg.shard", 9), cuda)
    metas = [dict(resize_shape=(32, 128, 3)) for _ in range(n)]
    with torch.no_grad():
        want = [r["text"] for r in m(img, [dict(mm) for mm in metas], return_loss=False)]
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        for mm in metas:
            mm["valid_ratio"] = 1.0

        def decode_local(lo, hi):
            feat = m.extract_feat(img[lo:hi], test=True)["output"]
            return m.decoder(feat, m.encoder(feat, metas[lo:hi]), None, metas[lo:hi], train_mode=False)
        with torch.no_grad():
            res = tdist.recognize_sharded(decode_local, n, m.label_convertor)
            rows = tdist.all_gather_rows(torch.arange(6, dtype=torch.float32, device=cuda).reshape(3, 2), 3)
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        assert [r["text"] for r in res] == want
        assert rows.is_cuda and torch.equal(rows.cpu(), torch.arange(6, dtype=torch.float32).reshape(3, 2))
    finally:
        dist.destroy_process_group()
