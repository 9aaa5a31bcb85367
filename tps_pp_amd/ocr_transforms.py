"""GPU-side test pipeline of the recogniser: ResizeOCR + ToTensorOCR + NormalizeOCR on a batch (SURVEY.md section 8f,
row F4).

Mirror of `mmocr/datasets/pipelines/ocr_transforms.py:18-156` (reference): `ResizeOCR` keeps the constructor, its
assertions and the per-image host logic (`plan`: resized width, padded width, `valid_ratio`, `resize_shape`,
`pad_shape` exactly as `__call__` computes them, :83-121); the pixel work of the three transforms is ONE HIP kernel
over the whole batch (`tpspp_resize_normalize_fwd`) instead of a per-sample OpenCV / torchvision call in a CPU data
loader.  `OCRBatchPreprocessor` strings them together the way `configs/_base_/recog_pipelines/crnn_pp_pipeline.py:85-95`
does and returns the tensor plus the `img_metas` the recogniser reads (`resize_shape`, `valid_ratio`, ...).

`backend` is honoured as the reference forwards it to `mmcv.imresize` (ocr_transforms.py:34-36,46,65,99-101):
`'pillow'` -> Pillow's `Image.resize(size, Image.BILINEAR)` arithmetic, PINNED bit for bit against the installed Pillow
(tests/golden/resize_pillow.npz); `None` / `'cv2'` -> OpenCV's 8-bit INTER_LINEAR arithmetic, which could not be pinned
against OpenCV itself (not installed at build time): see oracle/resize_oracle.py and DESIGN.md section 4f; anything else
raises `ValueError` when the transform is applied, as `mmcv.imresize` does.  No CPU fallback.
"""
import math

import numpy as np
import torch

from . import _lib, ops
from .registry import Registry

PIPELINES = Registry("pipeline")


def _is_none_or_type(x, t):
    return x is None or isinstance(x, t)


@PIPELINES.register_module()
class ResizeOCR:
    """`ocr_transforms.py:18-129`: same arguments and assertions; `plan(img_shape)` is the host half of `__call__`."""

    def __init__(self, height, min_width=None, max_width=None, keep_aspect_ratio=True, img_pad_value=0,
                 width_downsample_ratio=1.0 / 16, backend=None):
        assert isinstance(height, (int, tuple))
        assert _is_none_or_type(min_width, (int, tuple))
        assert _is_none_or_type(max_width, (int, tuple))
        if not keep_aspect_ratio:
            assert max_width is not None, '"max_width" must assigned if "keep_aspect_ratio" is False'
        assert isinstance(img_pad_value, int)
        if isinstance(height, tuple):
            assert isinstance(min_width, tuple)
            assert isinstance(max_width, tuple)
            assert len(height) == len(min_width) == len(max_width)
        self.height, self.min_width, self.max_width = height, min_width, max_width
        self.keep_aspect_ratio = keep_aspect_ratio
        self.img_pad_value = img_pad_value
        self.width_downsample_ratio = width_downsample_ratio
        self.backend = backend

    def interpolation(self):
        """`backend` -> the kernel's interpolation code (`mmcv.imresize`: None = the global backend, cv2 by default;
        an unknown name raises ValueError there as well, at call time)."""
        if self.backend in (None, "cv2"):
            return ops.RESIZE_CV2
        if self.backend == "pillow":
            return ops.RESIZE_PILLOW
        raise ValueError(f"backend: {self.backend} is not supported for resize. Supported backends are 'cv2', 'pillow'")

    def _dst(self, rank=0):
        if isinstance(self.height, int):
            return self.height, self.min_width, self.max_width
        idx = rank % len(self.height)          # multi-scale: one (height, width) pair per rank (:76-82)
        return self.height[idx], self.min_width[idx], self.max_width[idx]

    def plan(self, img_shape, rank=0):
        """-> dict(height, resize_w, out_w, valid_ratio, resize_shape, pad_shape) for one image of `img_shape`."""
        dst_height, dst_min_width, dst_max_width = self._dst(rank)
        ori_height, ori_width = img_shape[:2]
        c = img_shape[2] if len(img_shape) > 2 else 1
        valid_ratio = 1.0
        if self.keep_aspect_ratio:
            new_width = math.ceil(float(dst_height) / ori_height * ori_width)
            width_divisor = int(1 / self.width_downsample_ratio)
            if new_width % width_divisor != 0:
                new_width = round(new_width / width_divisor) * width_divisor
            if dst_min_width is not None:
                new_width = max(dst_min_width, new_width)
            if dst_max_width is not None:
                valid_ratio = min(1.0, 1.0 * new_width / dst_max_width)
                resize_width = min(dst_max_width, new_width)
                out_width = dst_max_width if new_width < dst_max_width else resize_width
            else:
                resize_width = out_width = new_width
        else:
            resize_width = out_width = dst_max_width
        return dict(height=dst_height, resize_w=int(resize_width), out_w=int(out_width), valid_ratio=valid_ratio,
                    resize_shape=(dst_height, int(resize_width), c), pad_shape=(dst_height, int(out_width), c))


@PIPELINES.register_module()
class NormalizeOCR:
    """`ocr_transforms.py:145-156`; `table()` tabulates ToTensorOCR + NormalizeOCR for the 256 byte values with
    torch's own fp32 arithmetic (x / 255, then (x - mean) / std), so the kernel's table lookup is exact."""

    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def table(self, device):
        v = torch.arange(256, dtype=torch.float32).div(255)
        m = torch.as_tensor(self.mean, dtype=torch.float32).view(-1, 1)
        s = torch.as_tensor(self.std, dtype=torch.float32).view(-1, 1)
        return v.view(1, -1).repeat(m.shape[0], 1).sub_(m).div_(s).contiguous().to(device)


class OCRBatchPreprocessor:
    """ResizeOCR -> ToTensorOCR -> NormalizeOCR on a list of uint8 HWC images (numpy arrays or tensors, any sizes)
    -> (tensor (N, C, height, width) on `device`, img_metas).  All images of a batch share the padded width (the
    reference's `max_width` when it pads; otherwise they must agree)."""

    def __init__(self, resize, normalize, device="cuda"):
        self.resize, self.normalize, self.device = resize, normalize, torch.device(device)
        self._lut = None

    def __call__(self, imgs, rank=0):
        if self.device.type != "cuda":
            raise _lib.TpsppError("OCRBatchPreprocessor: the HIP path needs a GPU device (no CPU fallback)")
        arrs = []
        for im in imgs:
            a = im.cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            if a.ndim == 2:                   # a grayscale crop as mmcv.imread(color_type='grayscale') hands it over
                a = a[:, :, None]
            if a.dtype != np.uint8 or a.ndim != 3:
                raise TypeError("OCRBatchPreprocessor: images must be uint8 (H, W, C) or (H, W) arrays")
            arrs.append(np.ascontiguousarray(a))
        if not arrs:
            raise ValueError("OCRBatchPreprocessor: empty batch")
        interpolation = self.resize.interpolation()          # (raises for an unknown backend)
        C = arrs[0].shape[2]
        plans = [self.resize.plan(a.shape, rank) for a in arrs]
        H, W = plans[0]["height"], plans[0]["out_w"]
        if any(a.shape[2] != C for a in arrs) or any(p["out_w"] != W or p["height"] != H for p in plans):
            raise ValueError("OCRBatchPreprocessor: the images of a batch must agree on channels and padded size "
                             "(set max_width, as the reference's test pipeline does)")
        if self._lut is None or self._lut.device != self.device:
            self._lut = self.normalize.table(self.device)
        if self._lut.shape[0] != C:
            raise ValueError("OCRBatchPreprocessor: mean / std need one entry per channel")
        sizes = np.array([a.size for a in arrs], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        packed = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(self.device, non_blocking=True)
        meta = np.stack([np.array([a.shape[0] for a in arrs]), np.array([a.shape[1] for a in arrs]),
                         np.array([p["resize_w"] for p in plans])]).astype(np.int32)
        meta_d = torch.from_numpy(meta).to(self.device)
        offs_d = torch.from_numpy(offs).to(self.device)
        out = ops.resize_normalize(packed, offs_d, meta_d[0], meta_d[1], meta_d[2], self._lut,
                                   self.resize.img_pad_value, len(arrs), C, H, W, interpolation)
        metas = [dict(ori_shape=a.shape, img_shape=p["resize_shape"], resize_shape=p["resize_shape"],
                      pad_shape=p["pad_shape"], valid_ratio=p["valid_ratio"],
                      img_norm_cfg=dict(mean=self.normalize.mean, std=self.normalize.std))
                 for a, p in zip(arrs, plans)]
        return out, metas
