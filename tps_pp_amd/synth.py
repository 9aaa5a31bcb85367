"""Exactly reproducible synthetic tensors (inputs and weights).

Every value is ``k / 2**23 - 1`` with ``k`` the top 24 bits of a counter-based 64-bit
integer hash of (seed, tensor name, flat index), i.e. a dyadic rational in [-1, 1) that is exactly
representable in fp32.  No dependence on the numpy / torch RNG streams, so the GPU box, this
container and the golden-vector generator (tests/golden/make_golden.py) all see bit-identical data
without shipping the inputs.

This is workload plumbing (bench.py, tests, smoke()); it holds no part of the rectification path.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def dyadic(shape, name: str, seed: int = 0, scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """fp32 array of ``shape``; entries ``offset + scale * (k/2**23 - 1)``, k in [0, 2**24)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    base = (_fnv1a64(name) ^ ((seed * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(_splitmix64(idx ^ np.uint64(base)) + np.uint64(base))
    k = (h >> np.uint64(40)).astype(np.int64)  # top 24 bits
    v = (k.astype(np.float64) / float(1 << 23)) - 1.0
    out = (offset + scale * v).astype(np.float32)
    return out.reshape(shape)


def smooth_image(shape, name: str, seed: int = 0) -> np.ndarray:
    """Image-like (band-limited) fp32 tensor (N, C, H, W) in [-1, 1]: a sum of a few low-frequency
    separable cosines whose phases/amplitudes come from :func:`dyadic`.  Used where the comparison
    must not be dominated by white-noise sensitivity to a flipped ``floor()``."""
    n, c, h, w = shape
    k = 4
    par = dyadic((n, c, k, 4), name, seed).astype(np.float64)
    yy = (np.arange(h, dtype=np.float64) + 0.5) / h
    xx = (np.arange(w, dtype=np.float64) + 0.5) / w
    out = np.zeros((n, c, h, w), dtype=np.float64)
    for j in range(k):
        fx = (j + 1) * (1.0 + 0.5 * par[:, :, j, 0])[..., None, None]
        fy = (0.5 * j + 0.5) * (1.0 + 0.5 * par[:, :, j, 1])[..., None, None]
        ph = np.pi * par[:, :, j, 2][..., None, None]
        am = (0.25 * (1.0 + par[:, :, j, 3]) / (j + 1))[..., None, None]
        out += am * np.cos(2 * np.pi * (fx * xx[None, None, None, :] + fy * yy[None, None, :, None]) + ph)
    return np.clip(out, -1.0, 1.0).astype(np.float32)


def state_dict_like(shapes: dict, seed: int = 0, overrides=None) -> dict:
    """name -> fp32 array for every (name, shape) in ``shapes``.

    Weights (ndim >= 2) are scaled by 1/sqrt(fan_in); 1-D tensors (biases, norm affine terms) by
    0.1 unless a rule in ``overrides`` (callable(name, shape) -> (scale, offset) | None) says
    otherwise, e.g. BatchNorm ``running_var`` / ``weight`` need positive values."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        rule = overrides(name, shape) if overrides is not None else None
        if rule is not None:
            scale, offset = rule
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            scale, offset = 1.0 / np.sqrt(fan_in), 0.0
        else:
            scale, offset = 0.1, 0.0
        out[name] = dyadic(shape, name, seed, scale, offset)
    return out
