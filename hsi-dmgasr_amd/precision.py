"""Precision mode of the HIP path.

  "bf16" : bf16 activations and MFMA operands, fp32 accumulation / statistics / softmax / sampler state
           (the throughput mode BASELINE.json's headline config names);
  "fp32" : fp32 activations; every matrix operand is split into bf16 hi + lo and hi*hi + hi*lo + lo*hi is
           accumulated in fp32 on the same MFMA pipeline (~1e-5 relative; the fp32 parity gate);
  "fp16" : fp16 activations and MFMA operands (11-bit significands instead of bf16's 8: the same kernels at the same rate,
           stores saturate at +-65504), and the convolutions of the two high-resolution levels (Cout <= 128: bound by their
           staging transform, not by the matrix pipe) multiply by fp16 hi + lo WEIGHTS in two MFMA passes - the weight rounding
           is the one error of a 16-bit mode that is the same in every step of a chain, i.e. a bias, not noise
           (tests/precision_emul.py).  "fp16x1" / "fp16x2": the second pass nowhere / wherever a kernel
           takes it (A/B forms).
"""
import os

# Package default (modules built without precision=...): the fastest mode that stays within the reference's tolerance
# (1e-3 / 0.01 dB / 0.001 deg on its validation chain); HSIDM_PRECISION or set_default_precision() choose another.
_default = os.environ.get("HSIDM_PRECISION", "fp16")


MODES = ("bf16", "fp32", "fp16", "fp16x1", "fp16x2")


def set_default_precision(p):
    global _default
    if p not in MODES:
        raise ValueError("precision must be one of %s" % (MODES,))
    _default = p


def get_default_precision():
    return _default


def resolve_precision(p):
    p = _default if p is None else p
    if p not in MODES:
        raise ValueError("precision must be one of %s, got %r" % (MODES, p))
    return p


def is_16bit(p):
    """bf16 / fp16 family: the throughput modes (2-byte activations, the persistent kernels)."""
    return p != "fp32"


def _policy(exp, cout, cin, ksize):
    """Evaluates a policy expression such as "cout <= 128 and (cin <= 192 or ksize == 1)": names, integers, comparisons, and / or /
    not and parentheses only (parsed, never eval'ed)."""
    import ast
    import operator as op
    cmp = {ast.LtE: op.le, ast.Lt: op.lt, ast.GtE: op.ge, ast.Gt: op.gt, ast.Eq: op.eq, ast.NotEq: op.ne}
    env = {"cout": cout, "cin": cin, "ksize": ksize}

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.BoolOp):
            vals = [ev(v) for v in n.values]
            return all(vals) if isinstance(n.op, ast.And) else any(vals)
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, ast.Not):
            return not ev(n.operand)
        if isinstance(n, ast.Compare):
            left = ev(n.left)
            for o, c in zip(n.ops, n.comparators):
                right = ev(c)
                if type(o) not in cmp or not cmp[type(o)](left, right):
                    return False
                left = right
            return True
        if isinstance(n, ast.Name) and n.id in env:
            return env[n.id]
        if isinstance(n, ast.Constant) and isinstance(n.value, int):
            return n.value
        raise ValueError("HSIDM_WIDE_POLICY: unsupported expression element %s" % type(n).__name__)
    return bool(ev(ast.parse(exp, mode="eval")))


def wide_weights(p, cout, cin=0, ksize=3):
    """Does a convolution (cout x cin x ksize x ksize) carry hi + lo weights in mode p (second MFMA pass)?"""
    exp = os.environ.get("HSIDM_WIDE_POLICY")          # diagnostic (tools/policy_probe.py): comparisons over cout, cin, ksize
    if exp and p == "fp16":
        return _policy(exp, cout, cin, ksize)
    return p == "fp16x2" or (p == "fp16" and cout <= 128)
