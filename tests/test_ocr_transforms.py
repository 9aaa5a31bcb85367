"""SURVEY.md section 8f row F4 (GPU ResizeOCR + ToTensorOCR + NormalizeOCR).

CPU part: the host logic and the normalisation against the reference's own known answers
(tests/test_dataset/test_ocr_transforms.py:13-57 of the reference) -- for the oracle and for the mirror;
GPU part: the kernel bit for bit against the oracle on ragged batches.  The interpolation arithmetic itself is
"parity unpinned" against a real cv2 (OpenCV is not installed here: see oracle/resize_oracle.py); what stands in for
it are (i) expected arrays for tiny cases derived step by step from OpenCV's published 8-bit INTER_LINEAR arithmetic
(HAND_CASES below, tables in the comments) and (ii) properties any correct resize has.  Until a fixture produced by
cv2 itself exists THAT backend stays unpinned.

Round 6: `backend='pillow'` (ocr_transforms.py:34-36,46,65 -> mmcv.imresize -> Image.resize(size, Image.BILINEAR)) IS pinned:
tests/golden/resize_pillow.npz holds the installed Pillow's own outputs on seeded crops (tests/golden/make_resize_golden.py
regenerates the inputs from tps_pp_amd/synth.py); the oracle's restatement of Pillow's Resample.c and the HIP kernel both
reproduce them bit for bit."""
import math

import numpy as np
import pytest
import torch

from oracle import resize_oracle as RO
from tps_pp_amd import NormalizeOCR, OCRBatchPreprocessor, ResizeOCR, synth

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]      # crnn_pp_pipeline.py:1


def test_reference_known_answers_oracle_and_mirror():
    # test_resize_ocr of the reference: ones (64, 256, 3), height 32, widths 32..160
    img = np.ones((64, 256, 3), dtype=np.uint8)
    out, p = RO.resize_ocr(img, 32, 32, 160, True)
    m = ResizeOCR(32, min_width=32, max_width=160, keep_aspect_ratio=True).plan(img.shape)
    for q in (p, m):
        assert np.allclose([32, 160, 3], q["pad_shape"]) and math.isclose(q["valid_ratio"], 0.8)
        assert q["resize_w"] == 128 and tuple(q["resize_shape"]) == (32, 128, 3)
    assert out.shape == (32, 160, 3) and math.isclose(np.sum(out[:, 129:, :]), 0) and (out[:, :128] == 1).all()
    assert math.isclose(RO.resize_plan(img.shape, 32, 32, 160, False)["valid_ratio"], 1)
    assert math.isclose(ResizeOCR(32, min_width=32, max_width=160, keep_aspect_ratio=False).plan(img.shape)["valid_ratio"], 1)
    # test_normalize / test_to_tensor of the reference
    z = RO.to_tensor_normalize(np.zeros((10, 10, 3), dtype=np.uint8), [0.5] * 3, [0.5] * 3)
    assert np.allclose(z, -1)
    lut = NormalizeOCR([0.5] * 3, [0.5] * 3).table("cpu")
    assert lut.shape == (3, 256) and float(lut[0, 0]) == -1.0 and float(lut[2, 255]) == 1.0
    v = torch.arange(256, dtype=torch.float32).div(255)
    assert torch.equal(NormalizeOCR(MEAN, STD).table("cpu")[1], (v - 0.456) / 0.224)


def test_resize_plan_mirror_equals_oracle_on_many_shapes():
    for keep in (True, False):
        for h, w in [(19, 35), (25, 119), (64, 256), (31, 900), (48, 48), (7, 3), (32, 128), (100, 17)]:
            for mn, mx in [(32, 128), (32, 160), (None, 128), (48, 100)]:
                a = RO.resize_plan((h, w, 3), 32, mn, mx, keep)
                b = ResizeOCR(32, min_width=mn, max_width=mx, keep_aspect_ratio=keep).plan((h, w, 3))
                assert all(a[k] == b[k] for k in a), (keep, h, w, mn, mx, a, b)


def test_constructor_assertions():
    with pytest.raises(AssertionError):
        ResizeOCR(32.0)
    with pytest.raises(AssertionError):
        ResizeOCR(32, keep_aspect_ratio=False)                 # max_width missing
    with pytest.raises(AssertionError):
        ResizeOCR((32, 48), min_width=32, max_width=128)


def test_oracle_bilinear_properties():
    g = np.random.default_rng(5)
    img = g.integers(0, 256, (23, 57, 3), dtype=np.uint8)
    assert np.array_equal(RO.imresize_bilinear_u8(img, (57, 23)), img)                     # same size: copy
    c = np.full((9, 11, 3), 77, dtype=np.uint8)
    assert (RO.imresize_bilinear_u8(c, (40, 32)) == 77).all()                              # constants stay constant
    big = RO.imresize_bilinear_u8(img, (114, 46))
    assert big.shape == (46, 114, 3) and big.min() >= img.min() and big.max() <= img.max() # convex weights
    half = RO.imresize_bilinear_u8(big, (57, 23))                                          # exact 2x2 shrink: area mean
    s = big.astype(np.int32)
    assert np.array_equal(half, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2))


# ---- expected outputs derived by hand from cv::resize(INTER_LINEAR) on uint8 (modules/imgproc/src/resize.cpp) -------------
# For a destination index d:  f = float((d + 0.5) * scale - 0.5) with scale = 1 / (dst / src) in double;  s = floor(f);
# w = f - s;  in x only: s < 0 -> (s, w) = (0, 0);  s >= src - 1 -> (s, w) = (src - 1, 0);  11-bit coefficients
# a0 = saturate_cast<short>((1 - w) * 2048), a1 = saturate_cast<short>(w * 2048) (round to nearest even).  Horizontal pass
# (HResizeLinear) in int32:  S[y][d] = img[y][s] * a0 + img[y][s + 1] * a1.  Vertical pass (VResizeLinear, rows clamped to
# [0, H - 1], weights b0 / b1 kept):  out = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
# An exact 2 x 2 shrink takes INTER_AREA instead: (a + b + c + d + 2) >> 2.
HAND_CASES = {
    # 3 -> 5 columns: scale 0.6; f = -0.2, 0.4, 1.0, 1.6, 2.2 -> (s, a0, a1) = (0,2048,0) (0,1229,819) (1,2048,0) (1,819,1229)
    # (2,2048,0); rows unchanged (b = 2048, 0).  e.g. d = 1, row 0: S = 10*1229 + 20*819 = 28670; 28670 >> 4 = 1791;
    # (2048*1791) >> 16 = 55; (55 + 2) >> 2 = 14.   d = 3: S = 20*819 + 40*1229 = 65540 -> 4096 -> 128 -> 130 >> 2 = 32.
    "A: 2x3 -> 2x5": ([[10, 20, 40], [255, 0, 128]], (5, 2), [[10, 14, 20, 32, 40], [255, 153, 0, 77, 128]]),
    # 2 -> 3 rows: scale 2/3; f = -1/6, 1/2, 7/6 -> rows (0,0) (0,1) (1,1) with (b0, b1) = (341,1707) (1024,1024) (1707,341);
    # x unchanged.  Row 1: (100 + 0) / 2 = 50;  S = 50*2048 and 255*2048: ((1024*6400) >> 16) + ((1024*32640) >> 16) = 100 + 510
    # -> (610 + 2) >> 2 = 153.
    "B: 2x2 -> 3x2": ([[100, 50], [0, 255]], (2, 3), [[100, 50], [50, 153], [0, 255]]),
    # 4 -> 3 in both directions: scale 4/3; f = 1/6, 3/2, 17/6 -> (s, a0, a1) = (0,1707,341) (1,1024,1024) (2,341,1707).
    # Top-left: S0 = 0*1707 + 64*341 = 21824 -> 1364; S1 = 16*1707 + 80*341 = 54592 -> 3412;
    # ((1707*1364) >> 16) + ((341*3412) >> 16) = 35 + 17 = 52 -> 54 >> 2 = 13.
    "C: 4x4 -> 3x3": ([[0, 64, 128, 255], [16, 80, 144, 240], [32, 96, 160, 224], [48, 112, 176, 208]], (3, 3),
                      [[13, 99, 232], [35, 120, 219], [56, 141, 204]]),
    # 2 -> 7 columns (f = -0.357.. -0.071.. 0.214 0.5 0.786 1.071 1.357: the ends clamp) and 3 -> 4 rows
    # (f = -0.125 0.625 1.375 2.125: (b0, b1) = (256,1792) (768,1280) (1280,768) (1792,256), rows (0,0) (0,1) (1,2) (2,2)).
    "E: 3x2 -> 4x7": ([[0, 255], [255, 0], [7, 9]], (7, 4),
                      [[0, 0, 55, 127, 200, 255, 255], [159, 159, 145, 127, 109, 96, 96], [162, 162, 128, 83, 37, 3, 3],
                       [7, 7, 7, 8, 8, 9, 9]]),
    # exact 2 x 2 shrink -> INTER_AREA: (1+3+9+11+2) >> 2 = 6, (5+7+13+15+2) >> 2 = 10, (250+251+254+255+2) >> 2 = 253
    "D: 2x6 -> 1x3 (area)": ([[1, 3, 5, 7, 250, 251], [9, 11, 13, 15, 254, 255]], (3, 1), [[6, 10, 253]]),
}


def test_hand_derived_opencv_vectors_pin_the_oracle():
    for name, (src, size, want) in HAND_CASES.items():
        img = np.array(src, dtype=np.uint8)[:, :, None]
        got = RO.imresize_bilinear_u8(img, size)[:, :, 0]
        assert np.array_equal(got, np.array(want, dtype=np.uint8)), (name, got.tolist())
        rgb = np.repeat(img, 3, axis=2)                        # the channels are independent
        assert np.array_equal(RO.imresize_bilinear_u8(rgb, size), np.repeat(np.array(want, np.uint8)[:, :, None], 3, 2))


def test_oracle_resize_properties():
    """What any correct INTER_LINEAR resize does, whatever its fixed-point details."""
    ramp = np.tile(np.arange(0, 250, 5, dtype=np.uint8)[None, :, None], (6, 1, 1))            # 6 x 50, monotone in x
    for size in ((128, 32), (37, 6), (50, 19), (200, 3)):
        r = RO.imresize_bilinear_u8(ramp, size)[:, :, 0].astype(np.int32)
        assert (np.diff(r, axis=1) >= 0).all(), f"monotone ramp, size {size}"
        # (two truncating products in the vertical pass: rows of a column-only pattern may differ by one level)
        assert np.abs(r - r[:1]).max() <= 1, "rows of a column-only pattern stay equal up to one level"
        assert r.min() >= 0 and r.max() <= 245
    for v in (0, 1, 127, 254, 255):
        c = np.full((5, 9, 2), v, dtype=np.uint8)
        for size in ((9, 5), (31, 17), (4, 2), (18, 10)):
            assert (RO.imresize_bilinear_u8(c, size) == v).all(), (v, size)
    # ResizeOCR pads on the right with img_pad_value, never on the left, and only beyond the resized width
    img = np.full((16, 40, 1), 200, dtype=np.uint8)
    out, plan = RO.resize_ocr(img, 32, min_width=32, max_width=128, keep_aspect_ratio=True, img_pad_value=9)
    assert plan["resize_w"] == 80 and out.shape == (32, 128, 1)
    assert (out[:, :80] == 200).all() and (out[:, 80:] == 9).all()


@pytest.mark.gpu
def test_gpu_kernel_reproduces_the_hand_derived_vectors(cuda):
    for name, (src, size, want) in HAND_CASES.items():
        img = np.array(src, dtype=np.uint8)[:, :, None]
        w, h = size
        pre = OCRBatchPreprocessor(ResizeOCR(h, min_width=w, max_width=w, keep_aspect_ratio=False), NormalizeOCR([0.0], [1.0]), cuda)
        out, _ = pre([img])
        ref = RO.to_tensor_normalize(np.array(want, dtype=np.uint8)[:, :, None], [0.0], [1.0])
        assert np.array_equal(out.cpu().numpy()[0].view(np.uint32), ref.view(np.uint32)), name
    # properties on the device: constants, identity at scale 1, monotone ramps
    c = np.full((5, 9, 1), 131, dtype=np.uint8)
    pre = OCRBatchPreprocessor(ResizeOCR(32, min_width=128, max_width=128, keep_aspect_ratio=False), NormalizeOCR([0.0], [1.0]), cuda)
    assert (pre([c])[0].cpu().numpy() == np.float32(131) / np.float32(255)).all()
    ramp = np.tile(np.arange(0, 250, 5, dtype=np.uint8)[None, :, None], (6, 1, 1))
    r = pre([ramp])[0].cpu().numpy()[0, 0]
    assert (np.diff(r, axis=1) >= 0).all() and np.abs(r - r[:1]).max() <= np.float32(1.001 / 255)
    same = np.random.default_rng(3).integers(0, 256, (32, 128, 1), dtype=np.uint8)
    assert np.array_equal(pre([same])[0].cpu().numpy()[0, 0], same[:, :, 0].astype(np.float32) / np.float32(255))


def ragged_images(n, seed):
    g = np.random.default_rng(seed)
    shapes = [(64, 256), (19, 35), (25, 119), (32, 128), (31, 400), (48, 48), (7, 3), (100, 17), (33, 77), (16, 64)]
    return [np.ascontiguousarray((synth.dyadic((h, w, 3), f"ocr.{seed}.{i}", seed) * 127.5 + 127.5).astype(np.uint8)
                                 if i % 2 else g.integers(0, 256, (h, w, 3), dtype=np.uint8))
            for i, (h, w) in enumerate((shapes * ((n + 9) // 10))[:n])]


@pytest.mark.gpu
@pytest.mark.parametrize("keep,mn,mx,pad", [(False, 32, 128, 0), (True, 32, 128, 0), (True, 32, 160, 7), (False, None, 100, 255)])
def test_gpu_batch_preprocessor_equals_oracle(cuda, keep, mn, mx, pad):
    imgs = ragged_images(23, 3)
    pre = OCRBatchPreprocessor(ResizeOCR(32, min_width=mn, max_width=mx, keep_aspect_ratio=keep, img_pad_value=pad),
                               NormalizeOCR(MEAN, STD), cuda)
    out, metas = pre(imgs)
    ref, plans = RO.preprocess_batch(imgs, 32, mn, mx, keep, pad, MEAN, STD)
    assert out.dtype == torch.float32 and tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    for m, p in zip(metas, plans):
        assert m["valid_ratio"] == p["valid_ratio"] and tuple(m["resize_shape"]) == tuple(p["resize_shape"])
        assert tuple(m["pad_shape"]) == tuple(p["pad_shape"])


@pytest.mark.gpu
def test_gpu_preprocessor_feeds_the_recogniser(cuda):
    """Crops -> GPU preprocessing -> recogniser, equal to the oracle's preprocessing fed to the same recogniser."""
    from test_gpu_head import build_recognizer
    m = build_recognizer(cuda)
    imgs = ragged_images(6, 8)
    pre = OCRBatchPreprocessor(ResizeOCR(32, min_width=32, max_width=128, keep_aspect_ratio=False),
                               NormalizeOCR(MEAN, STD), cuda)
    x, metas = pre(imgs)
    ref, _ = RO.preprocess_batch(imgs, 32, 32, 128, False, 0, MEAN, STD)
    with torch.no_grad():
        a = m(x, metas, return_loss=False)
        b = m(torch.from_numpy(ref).to(cuda), [dict(mm) for mm in metas], return_loss=False)
    assert [r["text"] for r in a] == [r["text"] for r in b]
    with pytest.raises(ValueError):
        OCRBatchPreprocessor(ResizeOCR(32, min_width=32, max_width=None), NormalizeOCR(MEAN, STD), cuda)(imgs)


@pytest.mark.gpu
def test_gpu_preprocessor_gray_and_upscaled_images(cuda):
    """One-channel crops, crops smaller than the target (up-scaling in both directions) and a single-row crop."""
    g = np.random.default_rng(11)
    imgs = [g.integers(0, 256, (h, w, 1), dtype=np.uint8) for h, w in [(8, 20), (3, 5), (1, 9), (40, 300), (32, 128), (64, 256)]]
    pre = OCRBatchPreprocessor(ResizeOCR(32, min_width=32, max_width=128, keep_aspect_ratio=True, img_pad_value=3),
                               NormalizeOCR([0.5], [0.25]), cuda)
    out, metas = pre(imgs)
    ref, plans = RO.preprocess_batch(imgs, 32, 32, 128, True, 3, [0.5], [0.25])
    assert tuple(out.shape) == (6, 1, 32, 128)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert [m["valid_ratio"] for m in metas] == [p["valid_ratio"] for p in plans]
    with pytest.raises(TypeError):
        pre([np.zeros((4, 4, 1), dtype=np.float32)])
    with pytest.raises(ValueError):
        pre([])


# ---- backend='pillow': pinned against the installed Pillow's own outputs (tests/golden/resize_pillow.npz) -------------------
def pillow_fixture():
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if here not in sys.path:
        sys.path.insert(0, here)
    import make_resize_golden as MG
    G = np.load(os.path.join(here, "resize_pillow.npz"))
    cases = []
    for i, ((H, W, C, w, h), kind) in enumerate(zip(G["cases"].tolist(), G["kinds"].tolist())):
        assert (H, W, C, w, h, kind) == tuple(MG.CASES[i]), "fixture and generator disagree: regenerate the fixture"
        cases.append((MG.make_input(i, H, W, C, kind), (w, h), G[f"out{i}"]))
    return cases


def test_pillow_oracle_reproduces_the_pillow_fixture():
    cases = pillow_fixture()
    assert len(cases) >= 12
    for i, (img, size, want) in enumerate(cases):
        got = RO.imresize_pillow_bilinear_u8(img, size)
        assert got.shape == want.shape and np.array_equal(got, want), (i, img.shape, size)
    # and, where Pillow is importable (the build container), against the library itself on fresh shapes
    try:
        from PIL import Image
    except ImportError:
        return
    g = np.random.default_rng(9)
    for t in range(40):
        H, W, w = int(g.integers(1, 80)), int(g.integers(1, 250)), int(g.integers(4, 161))
        img = g.integers(0, 256, (H, W, 3), dtype=np.uint8)
        assert np.array_equal(RO.imresize_pillow_bilinear_u8(img, (w, 32)), np.array(Image.fromarray(img).resize((w, 32), Image.BILINEAR)))


def test_backend_is_honoured_or_refused():
    assert ResizeOCR(32, max_width=128).interpolation() == 0
    assert ResizeOCR(32, max_width=128, backend="cv2").interpolation() == 0
    assert ResizeOCR(32, max_width=128, backend="pillow").interpolation() == 1
    with pytest.raises(ValueError, match="not supported for resize"):
        ResizeOCR(32, max_width=128, backend="turbojpeg").interpolation()      # (mmcv.imresize raises ValueError as well)
    # the two backends are different arithmetic: on a 3.1x horizontal shrink Pillow averages 7 source columns, OpenCV takes 2
    img = (synth.dyadic((31, 400, 3), "backend.diff", 1) * 127.5 + 127.5).astype(np.uint8)
    a, _ = RO.resize_ocr(img, 32, 32, 128, False, backend="pillow")
    b, _ = RO.resize_ocr(img, 32, 32, 128, False)
    assert a.shape == b.shape and not np.array_equal(a, b)


@pytest.mark.gpu
def test_gpu_pillow_backend_equals_the_pillow_fixture_bit_for_bit(cuda):
    """`ResizeOCR(backend='pillow')` on the GPU against the installed Pillow's outputs: every fixture case alone (so that its
    own width is the batch's width) through `OCRBatchPreprocessor`, identity normalisation -> value / 255 exactly."""
    for i, (img, (w, h), want) in enumerate(pillow_fixture()):
        C = img.shape[2]
        pre = OCRBatchPreprocessor(ResizeOCR(h, min_width=w, max_width=w, keep_aspect_ratio=False, backend="pillow"),
                                   NormalizeOCR([0.0] * C, [1.0] * C), cuda)
        out, _ = pre([img])
        ref = RO.to_tensor_normalize(want, [0.0] * C, [1.0] * C)
        got = out.cpu().numpy()[0]
        assert got.shape == ref.shape, (i, got.shape, ref.shape)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (i, img.shape, (w, h), float(np.abs(got - ref).max() * 255))
    with pytest.raises(ValueError, match="not supported for resize"):
        OCRBatchPreprocessor(ResizeOCR(32, max_width=128, keep_aspect_ratio=False, backend="nope"), NormalizeOCR([0.0], [1.0]), cuda)(
            [np.zeros((4, 4, 1), np.uint8)])


@pytest.mark.gpu
@pytest.mark.parametrize("keep,mn,mx,pad", [(False, 32, 128, 0), (True, 32, 128, 0), (True, 32, 160, 7), (False, None, 100, 255)])
def test_gpu_pillow_backend_ragged_batches_equal_the_pinned_oracle(cuda, keep, mn, mx, pad):
    imgs = ragged_images(23, 5) + [np.random.default_rng(2).integers(0, 256, (150, 1000, 3), dtype=np.uint8)]
    pre = OCRBatchPreprocessor(ResizeOCR(32, min_width=mn, max_width=mx, keep_aspect_ratio=keep, img_pad_value=pad, backend="pillow"),
                               NormalizeOCR(MEAN, STD), cuda)
    out, metas = pre(imgs)
    ref, plans = RO.preprocess_batch(imgs, 32, mn, mx, keep, pad, MEAN, STD, backend="pillow")
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    cv, _ = RO.preprocess_batch(imgs, 32, mn, mx, keep, pad, MEAN, STD)
    assert not np.array_equal(ref, cv)                                        # (the other backend is different arithmetic)
    for m, p in zip(metas, plans):
        assert m["valid_ratio"] == p["valid_ratio"] and tuple(m["resize_shape"]) == tuple(p["resize_shape"])
    # grayscale crops handed over as (H, W) arrays, as mmcv.imread(color_type='grayscale') does
    gray = [im[:, :, 0].copy() for im in imgs[:7]]
    pre1 = OCRBatchPreprocessor(ResizeOCR(32, min_width=mn, max_width=mx, keep_aspect_ratio=keep, img_pad_value=pad, backend="pillow"),
                                NormalizeOCR([0.5], [0.5]), cuda)
    ref1, _ = RO.preprocess_batch([g[:, :, None] for g in gray], 32, mn, mx, keep, pad, [0.5], [0.5], backend="pillow")
    assert np.array_equal(pre1(gray)[0].cpu().numpy().view(np.uint32), ref1.view(np.uint32))
