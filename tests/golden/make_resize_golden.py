"""Writes tests/golden/resize_pillow.npz: the INSTALLED Pillow's own outputs of

    np.array(Image.fromarray(img).resize((w, h), Image.BILINEAR))          # = mmcv.imresize(img, (w, h), backend='pillow')

(the call ResizeOCR makes with backend='pillow': mmocr/datasets/pipelines/ocr_transforms.py:46,65,99-101 -> mmcv/image/
geometric.py) on seeded uint8 crops.  Inputs are regenerated from tps_pp_amd/synth.py by the tests (never stored); only
Pillow's OUTPUTS are committed.  Every time it runs, the script also asserts that oracle/resize_oracle.py's restatement of
Pillow's Resample.c reproduces every output bit for bit, on the fixture cases and on a few hundred random shapes.

    python tests/golden/make_resize_golden.py            (needs Pillow; run in the build container)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import resize_oracle as RO          # noqa: E402
from tps_pp_amd import synth                    # noqa: E402

# (source height, source width, channels, target width, target height, kind)
CASES = [
    (64, 256, 3, 128, 32, "noise"),       # the reference's own test shape (test_ocr_transforms.py:13-30): exact 2x shrink
    (19, 35, 3, 64, 32, "noise"),         # up-scaling in both directions
    (25, 119, 3, 152, 32, "smooth"),      # up in y, up in x, width not a multiple of 4 at the source
    (31, 400, 3, 128, 32, "noise"),       # down 3.1x in x (7 taps), up in y
    (100, 17, 1, 16, 32, "noise"),        # down 3.1x in y, tiny width, one channel
    (48, 48, 1, 32, 32, "binary"),        # 1.5x down both ways, saturated values
    (7, 3, 3, 16, 32, "noise"),           # strong up-scaling from a 7 x 3 crop
    (33, 77, 3, 77, 32, "noise"),         # only the vertical pass runs (width unchanged)
    (32, 300, 3, 128, 32, "smooth"),      # only the horizontal pass runs (height unchanged)
    (32, 128, 3, 128, 32, "noise"),       # no resize at all: a copy
    (150, 1000, 3, 160, 32, "noise"),     # 6.25x / 4.7x down: 13 and 11 taps (coefficients beyond the register cache)
    (1, 9, 1, 100, 32, "noise"),          # a single-row crop
    (40, 301, 1, 97, 32, "smooth"),       # odd sizes, one channel
    (57, 23, 3, 48, 32, "binary"),
]


def make_input(i, H, W, C, kind):
    v = synth.dyadic((H, W, C), f"resize.pillow.{i}", 7)
    if kind == "smooth":
        v = synth.smooth_image((1, C, H, W), f"resize.pillow.s{i}", 7)[0].transpose(1, 2, 0)
    img = np.clip(np.floor(v.astype(np.float64) * 128.0 + 128.0), 0, 255).astype(np.uint8)
    if kind == "binary":
        img = np.where(img >= 128, 255, 0).astype(np.uint8)
    return np.ascontiguousarray(img)


def main():
    import PIL
    from PIL import Image
    out = {}
    for i, (H, W, C, w, h, kind) in enumerate(CASES):
        img = make_input(i, H, W, C, kind)
        a = img[:, :, 0] if C == 1 else img
        ref = np.array(Image.fromarray(a).resize((w, h), Image.BILINEAR))
        if C == 1:
            ref = ref[:, :, None]
        got = RO.imresize_pillow_bilinear_u8(img, (w, h))
        assert np.array_equal(got, ref), f"oracle != Pillow on case {i} {CASES[i]}"
        out[f"out{i}"] = ref
    g = np.random.default_rng(0)
    for t in range(300):
        H, W = int(g.integers(1, 90)), int(g.integers(1, 300))
        h, w = (32, int(g.integers(8, 161))) if t % 3 else (int(g.integers(1, 70)), int(g.integers(1, 200)))
        C = (1, 3)[t % 2]
        img = g.integers(0, 256, (H, W, C), dtype=np.uint8)
        a = img[:, :, 0] if C == 1 else img
        ref = np.array(Image.fromarray(a).resize((w, h), Image.BILINEAR))
        assert np.array_equal(RO.imresize_pillow_bilinear_u8(a, (w, h)), ref), (H, W, C, h, w)
    out["cases"] = np.array([c[:5] for c in CASES], dtype=np.int32)
    out["kinds"] = np.array([c[5] for c in CASES])
    out["pillow_version"] = np.array(PIL.__version__)
    path = os.path.join(HERE, "resize_pillow.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(CASES)} cases, Pillow {PIL.__version__}; oracle == Pillow on them and on 300 random shapes")


if __name__ == "__main__":
    main()
