"""Quality indices of super-resolved cubes on the device (reference eval_hsi.py:27-121, used in sr_gae.py:468-474).

``quality_indices(truth, pred)`` takes the NCHW fp32 cubes the pipeline already holds on the GPU and returns one row per
cube: MPSNR, SAM (degrees), ERGAS, CC, RMSE, MSSIM - the indices of the reference's quality_assessment (eval_hsi.py:217-238),
which computes them on the host with numpy / skimage (SAM as a Python double loop over pixels).  Kernels: csrc/metrics.hip
through hsidm_hsi_metrics and hsidm_hsi_mssim.
"""
import torch

from . import _lib

NAMES = ("mpsnr", "sam", "ergas", "cc", "rmse", "mssim")


def quality_indices(truth, pred, ratio=4, data_range=1.0):
    """truth, pred: [P, C, H, W] fp32 device tensors (same shape).  Returns a [P, 6] fp32 device tensor, columns NAMES
    (MSSIM is NaN for cubes smaller than its 7x7 window)."""
    if truth.shape != pred.shape or truth.dim() != 4:
        raise ValueError("expected two [P, C, H, W] tensors of equal shape, got %s and %s" % (tuple(truth.shape), tuple(pred.shape)))
    truth = truth.to(torch.float32).contiguous()
    pred = pred.to(torch.float32).contiguous()
    P, C, H, W = truth.shape
    L = _lib.lib()
    nbytes = L.hsidm_hsi_metrics_workspace_bytes(P, C, H * W)
    if nbytes <= 0:
        _lib.check(nbytes, "hsi_metrics_workspace_bytes")
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=truth.device)
    out = torch.empty((P, 5), dtype=torch.float32, device=truth.device)
    _lib.check(L.hsidm_hsi_metrics(_lib.ptr(truth), _lib.ptr(pred), P, C, H * W, float(ratio), float(data_range),
                                   _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr()), "hsi_metrics")
    ss = torch.full((P, 1), float("nan"), dtype=torch.float32, device=truth.device)
    if H >= 7 and W >= 7:
        nb = L.hsidm_hsi_mssim_workspace_bytes(P, C, H, W)
        if nb <= 0:
            _lib.check(nb, "hsi_mssim_workspace_bytes")
        ws2 = torch.empty(nb // 8, dtype=torch.float64, device=truth.device)
        _lib.check(L.hsidm_hsi_mssim(_lib.ptr(truth), _lib.ptr(pred), P, C, H, W, float(data_range), _lib.ptr(ws2), _lib.ptr(ss),
                                     _lib.stream_ptr()), "hsi_mssim")
    return torch.cat([out, ss], dim=1)


def as_dicts(table):
    """[P, 6] tensor -> list of {name: float} (one host synchronisation)."""
    rows = table.detach().cpu().tolist()
    return [dict(zip(NAMES, r)) for r in rows]


def color_correction(guide, pred, num_channels=None):
    """eval_hsi.py:259-274 (caller sr_gae.py:340): match each band of pred [P, C, H, W] to the mean and (population)
    standard deviation of the same band of guide [P, C, h, w], clip to [0, 1].  Bands >= num_channels come back zero, as the
    reference leaves them; None corrects every band."""
    if guide.dim() != 4 or pred.dim() != 4 or guide.shape[:2] != pred.shape[:2]:
        raise ValueError("expected guide [P, C, h, w] and pred [P, C, H, W], got %s and %s" % (tuple(guide.shape), tuple(pred.shape)))
    guide = guide.to(torch.float32).contiguous()
    pred = pred.to(torch.float32).contiguous()
    P, C, H, W = pred.shape
    nch = C if num_channels is None else int(num_channels)
    if nch > C:
        raise IndexError("num_channels %d exceeds the cube's %d bands" % (nch, C))   # the reference indexes band c and raises
    L = _lib.lib()
    ws = torch.empty(L.hsidm_color_correction_workspace_bytes(P, C) // 8, dtype=torch.float64, device=pred.device)
    out = torch.empty_like(pred)
    _lib.check(L.hsidm_color_correction(_lib.ptr(guide), guide.shape[2] * guide.shape[3], _lib.ptr(pred), _lib.ptr(out), P, C, H * W, nch,
                                        _lib.ptr(ws), _lib.stream_ptr()), "color_correction")
    return out
