"""CPU oracle for SURVEY.md section 8f row F4: ResizeOCR + ToTensorOCR + NormalizeOCR.

TEST INFRASTRUCTURE, NOT PRODUCT: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py`` may import this.

PARITY UNPINNED for the interpolation.  The reference resizes with ``mmcv.imresize`` (backend cv2, 'bilinear')
= OpenCV ``cv::resize(..., INTER_LINEAR)`` on uint8 (mmocr/datasets/pipelines/ocr_transforms.py:101-121).  Neither
cv2 nor mmcv is installed here (third-party: opencv-python, unpinned by the reference; mmcv-full 1.3.8-1.5.0), so
``imresize_bilinear_u8`` restates OpenCV's published 8-bit algorithm (modules/imgproc/src/resize.cpp: 11-bit
fixed-point coefficients, INTER_RESIZE_COEF_BITS = 11; the INTER_AREA substitution for an exact 2x2 shrink) and could
not be checked against OpenCV itself.  What IS pinned, against the reference's own tests
(tests/test_dataset/test_ocr_transforms.py:13-57): the host logic of ResizeOCR.__call__ (widths, padding,
valid_ratio, shapes), ToTensorOCR and NormalizeOCR.
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


def resize_ocr(img, height, min_width=None, max_width=None, keep_aspect_ratio=True, img_pad_value=0):
    """ResizeOCR on one uint8 HWC image -> (padded image, plan)."""
    p = resize_plan(img.shape, height, min_width, max_width, keep_aspect_ratio)
    r = imresize_bilinear_u8(img, (p["resize_w"], height))
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


def preprocess_batch(imgs, height, min_width, max_width, keep_aspect_ratio, img_pad_value, mean, std):
    """The test pipeline of configs/_base_/recog_pipelines/crnn_pp_pipeline.py:85-95 on a list of images ->
    (N, C, height, max_width) fp32, list of plans."""
    outs, plans = [], []
    for im in imgs:
        r, p = resize_ocr(im, height, min_width, max_width, keep_aspect_ratio, img_pad_value)
        outs.append(to_tensor_normalize(r, mean, std))
        plans.append(p)
    return np.stack(outs), plans
