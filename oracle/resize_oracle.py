"""CPU oracle for SURVEY.md section 8f row F4: ResizeOCR + ToTensorOCR + NormalizeOCR.

TEST INFRASTRUCTURE, NOT PRODUCT: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py`` may import this.

The reference resizes with ``mmcv.imresize(img, size, backend=self.backend)`` (mmocr/datasets/pipelines/ocr_transforms.py:
34-36,46,65,99-121); mmcv is a third-party dependency absent from /root/reference (mmcv-full 1.3.8-1.5.0), so both of its
backends are restated here from the libraries it calls:

* backend 'pillow' -- ``Image.fromarray(img).resize(size, Image.BILINEAR)``: ``imresize_pillow_bilinear_u8`` restates Pillow's
  src/libImaging/Resample.c and is PINNED: tests/golden/make_resize_golden.py runs the installed Pillow (12.2) on seeded crops,
  commits its outputs (tests/golden/resize_pillow.npz) and asserts this function reproduces them and 300 random shapes bit for bit.
* backend None / 'cv2' -- OpenCV ``cv::resize(..., INTER_LINEAR)`` on uint8: PARITY UNPINNED.  cv2 is not installed here
  (third-party: opencv-python, unpinned by the reference), so ``imresize_bilinear_u8`` restates OpenCV's published 8-bit
  algorithm (modules/imgproc/src/resize.cpp: 11-bit fixed-point coefficients, INTER_RESIZE_COEF_BITS = 11; the INTER_AREA
  substitution for an exact 2x2 shrink) and could not be checked against OpenCV itself.

Pinned against the reference's own tests (tests/test_dataset/test_ocr_transforms.py:13-57) in both cases: the host logic of
ResizeOCR.__call__ (widths, padding, valid_ratio, shapes), ToTensorOCR and NormalizeOCR.
"""
import math

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _cv_round(v):
    """cvRound: nearest, ties to even (lrint)."""
    return np.rint(v)


def _coeffs(src, dst):
    """Per destination index: source index and the two 11-bit weights, as resize.cpp computes them for
    INTER_LINEAR (ksize = 2): fx from a double scale, cast to float, floor, clamp at both ends with the weight of
    the missing neighbour set to zero; weights = saturate_cast<short>(w * 2048)."""
    scale = 1.0 / (float(dst) / float(src))                     # double
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def imresize_bilinear_u8(img, size):
    """img (H, W, C) uint8 -> (h, w, C) uint8, size = (w, h) as mmcv.imresize takes it."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    w, h = int(size[0]), int(size[1])
    if (h, w) == (H, W):
        return img.copy()
    if H == 2 * h and W == 2 * w:                               # INTER_LINEAR -> INTER_AREA for an exact 2x2 shrink
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx = _coeffs(W, w)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= W - 1
    fx = np.where(hi, np.float32(0), fx)
    sx = np.where(hi, W - 1, sx)
    a0 = np.clip(_cv_round((np.float32(1) - fx) * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
    a1 = np.clip(_cv_round(fx * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
    sx1 = np.minimum(sx + 1, W - 1)
    sy, fy = _coeffs(H, h)
    b0 = np.clip(_cv_round((np.float32(1) - fy) * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
    b1 = np.clip(_cv_round(fy * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
    y0 = np.clip(sy, 0, H - 1)
    y1 = np.clip(sy + 1, 0, H - 1)                              # rows are clamped, their weights kept
    src = img.astype(np.int32)
    rows = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]       # (H, w, C) horizontal pass
    S0, S1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


# ---- backend='pillow': PIL.Image.resize(size, Image.BILINEAR) on uint8 -- PINNED against the installed Pillow ---------------
# (mmcv.imresize(img, size, backend='pillow') = np.array(Image.fromarray(img).resize(size, Image.BILINEAR)),
# mmcv/image/geometric.py; the reference forwards ResizeOCR's `backend` there: ocr_transforms.py:46,65,99-101.)
# Restates Pillow's src/libImaging/Resample.c (12.x; unchanged since 7.0): precompute_coeffs in double -- support = 1.0 x
# max(scale, 1), window [int(center - support + 0.5), int(center + support + 0.5)) clipped to the image, triangle weights
# at (x + xmin - center + 0.5) / max(scale, 1) summed in ascending order and divided by their sum --, normalize_coeffs_8bpc
# (int(0.5 + k * 2^22)), then the horizontal pass into a uint8 image, then the vertical pass, each
# clip8((2^21 + sum(pixel * k)) >> 22).  tests/golden/make_resize_golden.py writes Pillow's own outputs for seeded crops to
# tests/golden/resize_pillow.npz and asserts this function reproduces every one bit for bit.
PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_coeffs(in_size, out_size):
    """-> list of (xmin, int32 coefficient array) per output index (precompute_coeffs + normalize_coeffs_8bpc)."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)                     # (C cast: truncation toward zero)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws, ww = [], 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            w = 1.0 - t if t < 1.0 else 0.0
            ws.append(w)
            ww += w
        ks = []
        for w in ws:
            if ww != 0.0:
                w = w / ww
            ks.append(int(-0.5 + w * (1 << PIL_PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PIL_PRECISION_BITS)))
        out.append((xmin, np.asarray(ks, dtype=np.int64)))
    return out


def _pil_clip8(v):
    return np.clip(v >> PIL_PRECISION_BITS, 0, 255)


def imresize_pillow_bilinear_u8(img, size):
    """img (H, W, C) or (H, W) uint8 -> (h, w[, C]) uint8, size = (w, h): Pillow's Image.resize(size, Image.BILINEAR)."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim in (2, 3)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, C = img.shape
    w, h = int(size[0]), int(size[1])
    cur = img.astype(np.int64)
    if w != W:                                                  # horizontal pass first (ImagingResample: need_horizontal)
        nxt = np.empty((H, w, C), dtype=np.int64)
        for xx, (xmin, k) in enumerate(_pil_coeffs(W, w)):
            acc = (1 << (PIL_PRECISION_BITS - 1)) + (cur[:, xmin:xmin + len(k), :] * k[None, :, None]).sum(1)
            nxt[:, xx, :] = _pil_clip8(acc)
        cur = nxt
    if h != H:
        nxt = np.empty((h, cur.shape[1], C), dtype=np.int64)
        for yy, (ymin, k) in enumerate(_pil_coeffs(H, h)):
            acc = (1 << (PIL_PRECISION_BITS - 1)) + (cur[ymin:ymin + len(k), :, :] * k[:, None, None]).sum(0)
            nxt[yy] = _pil_clip8(acc)
        cur = nxt
    out = cur.astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def resize_plan(img_shape, height, min_width=None, max_width=None, keep_aspect_ratio=True,
                width_downsample_ratio=1.0 / 16):
    """Host logic of ResizeOCR.__call__ (ocr_transforms.py:83-121) for one image:
    -> dict(resize_w, out_w, valid_ratio, resize_shape, pad_shape)."""
    ori_h, ori_w = img_shape[:2]
    c = img_shape[2] if len(img_shape) > 2 else 1
    valid_ratio = 1.0
    if keep_aspect_ratio:
        new_w = math.ceil(float(height) / ori_h * ori_w)
        div = int(1 / width_downsample_ratio)
        if new_w % div != 0:
            new_w = round(new_w / div) * div
        if min_width is not None:
            new_w = max(min_width, new_w)
        if max_width is not None:
            valid_ratio = min(1.0, 1.0 * new_w / max_width)
            resize_w = min(max_width, new_w)
            out_w = max_width if new_w < max_width else resize_w
        else:
            resize_w = out_w = new_w
    else:
        resize_w = out_w = max_width
    return dict(resize_w=int(resize_w), out_w=int(out_w), valid_ratio=valid_ratio,
                resize_shape=(height, int(resize_w), c), pad_shape=(height, int(out_w), c))


def resize_ocr(img, height, min_width=None, max_width=None, keep_aspect_ratio=True, img_pad_value=0, backend=None):
    """ResizeOCR on one uint8 HWC image -> (padded image, plan).  backend None / 'cv2': the (unpinned) OpenCV restatement;
    'pillow': the (pinned) Pillow restatement."""
    p = resize_plan(img.shape, height, min_width, max_width, keep_aspect_ratio)
    resize = imresize_pillow_bilinear_u8 if backend == "pillow" else imresize_bilinear_u8
    r = resize(img, (p["resize_w"], height))
    if p["out_w"] > p["resize_w"]:
        pad = np.full((height, p["out_w"] - p["resize_w"], img.shape[2]), img_pad_value, dtype=np.uint8)
        r = np.concatenate([r, pad], axis=1)                    # mmcv.impad: right / bottom padding
    return r, p


def to_tensor_normalize(img_u8, mean, std):
    """ToTensorOCR + NormalizeOCR (ocr_transforms.py:136-156) through torch, exactly as torchvision composes
    them: HWC uint8 -> CHW float / 255, then (x - mean) / std in fp32."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return t.sub_(m).div_(s).numpy()


def preprocess_batch(imgs, height, min_width, max_width, keep_aspect_ratio, img_pad_value, mean, std, backend=None):
    """The test pipeline of configs/_base_/recog_pipelines/crnn_pp_pipeline.py:85-95 on a list of images ->
    (N, C, height, max_width) fp32, list of plans."""
    outs, plans = [], []
    for im in imgs:
        r, p = resize_ocr(im, height, min_width, max_width, keep_aspect_ratio, img_pad_value, backend)
        outs.append(to_tensor_normalize(r, mean, std))
        plans.append(p)
    return np.stack(outs), plans
