"""Patch preparation on the device: min-max normalisation, the MATLAB-compatible bicubic x1/n, xn pair that turns a
ground-truth cube into the network's inputs (reference HStest.py:37-60, HStrain.py:49-70, imsize.py:35-158) and the
training set's crop + 8-way augmentation (HStrain.py:51-82, utils.py:3-28).

The tap tables are host arithmetic (float64, once per length pair, cached on the device); the resampling itself runs in
csrc/degrade.hip through hsidm_resample_axis / hsidm_minmax_normalize.  Cubes are NCHW fp32 device tensors [P, C, H, W].
"""
import math

import numpy as np
import torch

from . import _lib

_tap_cache = {}


def _keys_cubic(t):
    a = np.abs(t)
    return np.where(a <= 1.0, (1.5 * a - 2.5) * a * a + 1.0, np.where(a <= 2.0, ((-0.5 * a + 2.5) * a - 4.0) * a + 2.0, 0.0))


def tap_tables(in_length, out_length):
    """(weights [out, P] float64, indices [out, P] int32) of one axis: bicubic taps, widened by 1/scale when shrinking
    (antialiasing), centred like MATLAB (u = x/scale + (1 - 1/scale)/2, 1-based x), mirrored at the edges, normalised
    to sum 1, all-zero columns dropped (imsize.py:35-60)."""
    scale = out_length / in_length
    stretch = 1.0 / scale if scale < 1.0 else 1.0
    width = 4.0 * stretch
    centre = np.arange(1, out_length + 1, dtype=np.float64) / scale + 0.5 * (1.0 - 1.0 / scale)
    first = np.floor(centre - width / 2.0)
    n = int(math.ceil(width)) + 2
    pos = (first[:, None] + np.arange(n)[None, :] - 1).astype(np.int32)
    wts = _keys_cubic((centre[:, None] - pos - 1.0) / stretch) / stretch
    wts = wts / wts.sum(axis=1, keepdims=True)
    period = 2 * in_length
    m = np.mod(pos, period)
    pos = np.where(m < in_length, m, period - 1 - m).astype(np.int32)
    keep = np.any(wts != 0.0, axis=0)
    return wts[:, keep], pos[:, keep]


def _device_taps(in_length, out_length, dev):
    key = (in_length, out_length, str(dev))
    if key not in _tap_cache:
        w, idx = tap_tables(in_length, out_length)
        _tap_cache[key] = (torch.tensor(w, dtype=torch.float32, device=dev).contiguous(),
                           torch.tensor(idx, dtype=torch.int32, device=dev).contiguous(), w.shape[1])
    return _tap_cache[key]


def imresize(x, out_hw, clamp01=False):
    """x [P, C, H, W] fp32 -> [P, C, out_h, out_w]; the axis with the smaller scale factor is resampled first
    (imsize.py:140-152)."""
    x = x.to(torch.float32).contiguous()
    P, C, H, W = x.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    order = [0, 1] if oh / H <= ow / W else [1, 0]
    L = _lib.lib()
    cur, h, w = x, H, W
    for n, axis in enumerate(order):
        last = n == 1
        if axis == 0:
            wt, idx, taps = _device_taps(h, oh, x.device)
            out = torch.empty((P, C, oh, w), dtype=torch.float32, device=x.device)
            _lib.check(L.hsidm_resample_axis(_lib.ptr(cur), _lib.ptr(out), P * C, h, oh, w, _lib.ptr(wt), _lib.ptr(idx), taps,
                                             int(clamp01 and last), _lib.stream_ptr()), "resample_axis")
            h = oh
        else:
            wt, idx, taps = _device_taps(w, ow, x.device)
            out = torch.empty((P, C, h, ow), dtype=torch.float32, device=x.device)
            _lib.check(L.hsidm_resample_axis(_lib.ptr(cur), _lib.ptr(out), P * C * h, w, ow, 1, _lib.ptr(wt), _lib.ptr(idx), taps,
                                             int(clamp01 and last), _lib.stream_ptr()), "resample_axis")
            w = ow
        cur = out
    return cur


def lr_pair(gt, n_scale=4):
    """gt [P, C, H, W] in [0, 1] -> (ms [P, C, H/n, W/n], lms [P, C, H, W]), both clamped to [0, 1] (HStest.py:43-60)."""
    H, W = gt.shape[2], gt.shape[3]
    ms = imresize(gt, (H // n_scale, W // n_scale))          # the reference clamps ms only after lms was made from it
    lms = imresize(ms, (H, W), clamp01=True)
    return ms.clamp_(0.0, 1.0), lms


def minmax_normalize(x):
    """(x - min) / (max - min) over each whole cube of x [P, ...] (HStest.py:37)."""
    x = x.to(torch.float32).contiguous()
    P = x.shape[0]
    L = _lib.lib()
    ws = torch.empty(L.hsidm_minmax_workspace_bytes(P) // 4, dtype=torch.float32, device=x.device)
    out = torch.empty_like(x)
    _lib.check(L.hsidm_minmax_normalize(_lib.ptr(x), _lib.ptr(out), P, x.numel() // P, _lib.ptr(ws), _lib.stream_ptr()), "minmax_normalize")
    return out


def augment(x, mode):
    """One of the 8 training augmentations (utils.py:3-28) on the spatial axes of x [..., H, W]: 0 identity, 1 flipud,
    2 rot90, 3 flipud(rot90), 4 rot180, 5 flipud(rot180), 6 rot270, 7 flipud(rot270); quarter turns swap H and W."""
    mode = int(mode)
    if not 0 <= mode <= 7:
        raise ValueError("augmentation mode %r: the reference defines modes 0..7 (utils.py:3-28)" % (mode,))
    x = x.to(torch.float32).contiguous()
    H, W = x.shape[-2], x.shape[-1]
    oh, ow = (W, H) if mode in (2, 3, 6, 7) else (H, W)
    out = torch.empty(x.shape[:-2] + (oh, ow), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsidm_augment(_lib.ptr(x), _lib.ptr(out), x.numel() // (H * W), H, W, mode, _lib.stream_ptr()), "augment")
    return out


def training_items(cubes, rows, cols, modes, n_scale=4, lr_size=32, normalize=True):
    """HSTrainingData.__getitem__ (HStrain.py:26-82) for a batch: cubes [P, C, H, W] raw; per item p a crop origin
    (rows[p], cols[p]) and an augmentation mode.  Returns {'HR', 'SR', 'LR'} as the reference's dict: HR = the
    gt_size = lr_size * n_scale crop, LR = its bicubic x1/n, SR = LR resized back, all three augmented the same way,
    LR and SR clamped to [0, 1] (the crop origins and modes are the caller's random draws, HStrain.py:53-55, 28-31)."""
    gt_all = minmax_normalize(cubes) if normalize else cubes.to(torch.float32).contiguous()
    size = lr_size * n_scale
    P, C, H, W = gt_all.shape
    if len(rows) != P or len(cols) != P or len(modes) != P:
        raise ValueError("training_items: one (row, col, mode) per cube")
    for r, c in zip(rows, cols):
        if not (0 <= r <= H - size and 0 <= c <= W - size):
            raise ValueError("crop origin (%d, %d) outside a %dx%d cube for a %d-pixel patch" % (r, c, H, W, size))
    gt = torch.stack([gt_all[p, :, rows[p]:rows[p] + size, cols[p]:cols[p] + size] for p in range(P)]).contiguous()
    ms = imresize(gt, (lr_size, lr_size))
    lms = imresize(ms, (size, size), clamp01=True)           # made from the unclamped LR, as HStrain.py:60-62 does
    ms = ms.clamp_(0.0, 1.0)
    out = {"HR": [], "SR": [], "LR": []}
    for p in range(P):
        out["HR"].append(augment(gt[p], modes[p]))
        out["SR"].append(augment(lms[p], modes[p]))
        out["LR"].append(augment(ms[p], modes[p]))
    return {k: torch.stack(v) for k, v in out.items()}
