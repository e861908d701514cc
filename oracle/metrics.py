"""Oracle: the two HSI quality indices the parity gate uses (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

MPSNR follows eval_hsi.py:110-121 (mean over bands of skimage's
peak_signal_noise_ratio = 10*log10(range^2 / MSE), restated because skimage is
not part of this image); SAM follows eval_hsi.py:47-65 (mean spectral angle in
degrees over pixels whose two spectra are non-zero).  Inputs are (H, W, C).
"""
import numpy as np


def mpsnr(x_true, x_pred, data_range=1.0):
    t = x_true.astype(np.float32)
    p = x_pred.astype(np.float32)
    vals = []
    for k in range(t.shape[2]):
        err = np.mean((t[:, :, k].astype(np.float64) - p[:, :, k].astype(np.float64)) ** 2)
        vals.append(10.0 * np.log10((data_range ** 2) / err))
    return float(np.mean(vals))


def sam_degrees(x_true, x_pred):
    t = x_true.astype(np.float32).reshape(-1, x_true.shape[2])
    p = x_pred.astype(np.float32).reshape(-1, x_pred.shape[2])
    nt = np.linalg.norm(t, axis=1)
    npd = np.linalg.norm(p, axis=1)
    ok = (nt != 0) & (npd != 0)
    cos = np.sum(t[ok] * p[ok], axis=1) / (nt[ok] * npd[ok])
    ang = np.arccos(cos)
    return float(np.sum(ang) / np.count_nonzero(ok) * 180.0 / np.pi)
