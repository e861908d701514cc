"""Oracle: MATLAB-compatible bicubic resize of the reference's data pipeline (numpy, float64 inside).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates imsize.py:35-60 (``contributions``: antialiased separable kernel taps with symmetric boundary extension) and
imsize.py:116-158 (``imresize`` with ``output_shape``: the two axes are resampled in order of increasing scale), as used
by HStest.py:40-45 / HStrain.py:53-63 (gt -> ms = gt/4 -> lms = ms*4, then clamp to [0, 1]).  Pinned against outputs of
the reference function itself (tests/golden/imresize.npz).
"""
import math

import numpy as np


def cubic(x):
    """Keys cubic convolution kernel, a = -0.5 (imsize.py:25-32)."""
    ax = np.abs(np.asarray(x, dtype=np.float64))
    near = (1.5 * ax - 2.5) * ax * ax + 1.0
    far = ((-0.5 * ax + 2.5) * ax - 4.0) * ax + 2.0
    return np.where(ax <= 1.0, near, np.where(ax <= 2.0, far, 0.0))


def taps(in_length, out_length, kernel_width=4.0):
    """(weights [out, P] float64, indices [out, P] int) of one axis (imsize.py:35-60).  Shrinking widens the kernel by
    1/scale (antialiasing); out-of-range taps mirror about the edges; columns that are zero for every output are dropped."""
    scale = out_length / in_length
    if scale < 1.0:
        width = kernel_width / scale
        h = lambda t: scale * cubic(scale * t)
    else:
        width = kernel_width
        h = cubic
    x = np.arange(1, out_length + 1, dtype=np.float64)
    u = x / scale + 0.5 * (1.0 - 1.0 / scale)
    left = np.floor(u - width / 2.0)
    count = int(math.ceil(width)) + 2
    idx = (left[:, None] + np.arange(count)[None, :] - 1).astype(np.int32)           # zero-based source positions
    w = h(u[:, None] - idx - 1.0)
    w = w / w.sum(axis=1, keepdims=True)
    mirror = np.concatenate([np.arange(in_length), np.arange(in_length - 1, -1, -1)])
    idx = mirror[np.mod(idx, mirror.size)]
    keep = np.any(w != 0.0, axis=0)
    return w[:, keep], idx[:, keep]


def imresize(img, output_shape):
    """img (H, W) or (H, W, C) -> float64 array of spatial size output_shape (bicubic, imsize.py:116-158)."""
    a = np.asarray(img)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    scales = [output_shape[k] / a.shape[k] for k in range(2)]
    out = a.astype(np.float64)
    for axis in np.argsort(np.array(scales), kind="quicksort"):
        w, idx = taps(a.shape[axis], output_shape[axis])
        if axis == 0:
            out = np.einsum("op,opwc->owc", w, out[idx])
        else:
            out = np.einsum("op,hopc->hoc", w, out[:, idx])
    return out[:, :, 0] if squeeze else out


def lr_pair(gt, n_scale):
    """HStest.py:40-60: ms = imresize(gt, /n_scale), lms = imresize(ms, gt size); both clamped to [0, 1] as float32."""
    h, w = gt.shape[:2]
    ms = imresize(gt, (h // n_scale, w // n_scale))
    lms = imresize(ms, (h, w))
    return np.clip(ms.astype(np.float32), 0.0, 1.0), np.clip(lms.astype(np.float32), 0.0, 1.0)
