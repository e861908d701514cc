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

The fp16 family carries a precision SCHEDULE along the reverse chain (step_precision below): a step whose update feeds the
denoiser's output error into the state with a gain of WIDE_STEP_GAIN or more runs in the "fp32" mode.  That gain is
|c_x0 * sqrt(1/acp_t - 1)| - the coefficient of eps in x_{t-1} (reference diffusion.py:142-149: posterior_mean_coef1 *
sqrt_recipm1_alphas_cumprod).  With the reference's cosine schedule it is 31.6 at the first step of EVERY chain (beta is clamped at
0.999 there, diffusion.py:46) and 1.50 / 0.83 / 0.58 / 0.45 ... at the following steps whatever the chain length (near t = T the
schedule's alpha-bar is ~ (T - t)^2): four steps per chain - four in the shipped 20-step validation chain, four in the
benchmark's thousand.  (tests/precision_emul.py, key `is`: on the 20-step chain the first step alone carries 18 % of the fp16
mode's deviation from the reference, the first two 21 %, the first six 39 %.)
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


WIDE_STEP_GAIN = 0.5


def step_precision(p, eps_gain):
    """Mode of ONE reverse step of a chain run in mode p, given the step's gain on the denoiser's output (see the module docstring)."""
    if p in ("fp16", "fp16x1", "fp16x2") and eps_gain >= WIDE_STEP_GAIN and not os.environ.get("HSIDM_NO_STEP_SCHEDULE"):
        return "fp32"
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
