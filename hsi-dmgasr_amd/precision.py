"""Precision mode of the HIP path.

  "bf16" : bf16 activations and MFMA operands, fp32 accumulation / statistics / softmax / sampler state
           (the throughput mode BASELINE.json's headline config names);
  "fp32" : fp32 activations; every matrix operand is split into bf16 hi + lo and hi*hi + hi*lo + lo*hi is
           accumulated in fp32 on the same MFMA pipeline (~1e-5 relative; the fp32 parity gate).
"""
import os

_default = os.environ.get("HSIDM_PRECISION", "bf16")


def set_default_precision(p):
    global _default
    if p not in ("bf16", "fp32"):
        raise ValueError("precision must be 'bf16' or 'fp32'")
    _default = p


def get_default_precision():
    return _default


def resolve_precision(p):
    p = _default if p is None else p
    if p not in ("bf16", "fp32"):
        raise ValueError("precision must be 'bf16' or 'fp32', got %r" % (p,))
    return p
