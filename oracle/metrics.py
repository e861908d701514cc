"""Oracle: the two HSI quality indices the parity gate uses (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

ERGAS, CC and RMSE (eval_hsi.py:27-44, 68-80, 99-107) are restated for the on-device indices and pinned
to values of the reference itself (tests/golden/metrics2.npz).  MPSNR follows eval_hsi.py:110-121 (mean over bands of skimage's
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


def _bands(x_true, x_pred):
    """img_2d_mat, eval_hsi.py:83-96: (H, W, C) -> (C, H*W) float32."""
    c = x_true.shape[2]
    return (x_true.astype(np.float32).reshape(-1, c).T.copy(), x_pred.astype(np.float32).reshape(-1, c).T.copy())


def ergas(x_true, x_pred, ratio):
    """eval_hsi.py:27-44: (100/ratio) * sqrt(mean_band(MSE_band / mean(true_band)^2))."""
    t, p = _bands(x_true, x_pred)
    acc = 0.0
    for i in range(t.shape[0]):
        acc += np.mean((t[i] - p[i]) ** 2) / (np.mean(t[i]) ** 2)
    return float((100.0 / ratio) * np.sqrt(acc / t.shape[0]))


def cross_correlation(x_true, x_pred):
    """eval_hsi.py:68-80: mean over bands of the Pearson correlation of the two band images."""
    t, p = _bands(x_true, x_pred)
    t = t - t.mean(axis=1, keepdims=True)
    p = p - p.mean(axis=1, keepdims=True)
    return float(np.mean(np.sum(t * p, axis=1) / np.sqrt(np.sum(t * t, axis=1) * np.sum(p * p, axis=1))))


def rmse(x_true, x_pred):
    """eval_hsi.py:99-107: Frobenius norm of the difference / sqrt(number of elements)."""
    d = x_true.astype(np.float32) - x_pred.astype(np.float32)
    return float(np.linalg.norm(d) / np.sqrt(d.size))


def mssim(x_true, x_pred, data_range=1.0):
    """eval_hsi.py:124-135: mean over bands of skimage.metrics.structural_similarity(im1, im2, data_range=...) with its
    defaults, restated from skimage's documented algorithm (skimage is not part of this image: unpinned, like MPSNR):
    7x7 uniform filter, sample covariance (n / (n - 1)), K1 = 0.01, K2 = 0.03, border of (7 - 1) / 2 pixels cropped."""
    from scipy.ndimage import uniform_filter
    t = x_true.astype(np.float64)
    p = x_pred.astype(np.float64)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    cov = 49.0 / 48.0
    vals = []
    for k in range(t.shape[2]):
        a, b = t[:, :, k], p[:, :, k]
        ux, uy = uniform_filter(a, 7), uniform_filter(b, 7)
        vx = cov * (uniform_filter(a * a, 7) - ux * ux)
        vy = cov * (uniform_filter(b * b, 7) - uy * uy)
        vxy = cov * (uniform_filter(a * b, 7) - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        vals.append(s[3:-3, 3:-3].mean())
    return float(np.mean(vals))
