"""Helpers for the GPU parity tests: build product modules with the synthetic weights of tests/golden/synth.py,
compare with the oracle, and log every measured error to gpurun_out/parity.jsonl (kept by gpurun)."""
import json
import os

import numpy as np
import torch

from helpers import rel_err
from synth import synth_param

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# tolerances (relative Frobenius error vs the fp32 CPU oracle)
TOL = {"fp32": 1e-3,    # north_star: 1e-3 relative for the fp32 path (measured ~1e-5)
       "fp16": 1e-3, "fp16x1": 1.5e-3, "fp16x2": 1e-3,     # fp16 modes: per op ~3e-4 (tests/test_gpu_anchor.py holds each kernel to 6e-4)
       "fp16d0": 1.5e-3, "fp16d1": 1.5e-3, "fp16d2": 1.5e-3, "fp16d3": 1.5e-3,   # the dithered one-pass sets of the "fp16" policy's chain steps: a
       # single forward sees weights up to 7/8 ulp off (plain rounding: 1/2) - per-op regression gates; the policy's claim is the chain's
       "bf16": 4e-2}    # bf16 operands: a regression bound per op (measured 2e-3 ... 9e-3), not a north-star claim


def fill_synth(module, prefix, seed=0):
    sd = {k: torch.from_numpy(synth_param(prefix + k, tuple(v.shape), seed)) for k, v in module.state_dict().items()
          if not k.startswith("_")}
    module.load_state_dict(sd, strict=False)
    return {k: v.clone() for k, v in sd.items()}


def log_err(name, precision, err, extra=None):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rec = {"test": name, "precision": precision, "rel_err": err}
    if extra:
        rec.update(extra)
    with open(os.path.join(ROOT, "gpurun_out", "parity.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")


def check(name, precision, got, want, tol=None):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    assert np.isfinite(got).all(), "%s: non-finite output" % name
    e = rel_err(got, want)
    log_err(name, precision, e)
    tol = TOL[precision] if tol is None else tol
    assert e < tol, "%s [%s]: rel err %.3e >= %.1e" % (name, precision, e, tol)
    return e


def assert_stats(slab, y, tag=None):
    """Statistics slab [B, nsplit, C, 2] of a convolution against (sum, sum of squares) of its stored NHWC output y.
    conv_v2 / conv_v3 / conv1x1_g may sum a residual-free conv's fp32 values BEFORE the bf16 rounding of the store: the two
    sides then differ by the sum of HW zero-mean rounding errors of at most 2^-9 |y| each, bounded here at 6 sigma (kernels
    that sum the stored values pass the same bound trivially)."""
    yf = y.float()
    want = torch.stack([yf.sum(dim=(1, 2)), (yf * yf).sum(dim=(1, 2))], dim=2)         # [B, C, 2]
    got = slab.sum(dim=1)
    n_px = yf.shape[1] * yf.shape[2]
    ymax = float(yf.abs().max())
    noise = 6.0 * n_px ** 0.5 * 2.0 ** -9 * ymax / 3 ** 0.5
    err = (got - want).abs()
    lim0 = 2e-2 + noise + 2e-3 * want[..., 0].abs()
    lim1 = 2e-2 + 2.0 * ymax * noise + 2e-3 * want[..., 1].abs()
    assert bool((err[..., 0] <= lim0).all()), (tag, float(err[..., 0].max()))
    assert bool((err[..., 1] <= lim1).all()), (tag, float(err[..., 1].max()))
