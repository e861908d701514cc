"""Precision mode of the HIP path.

PUBLIC modes (both meet the reference's tolerance - 1e-3 relative, 0.01 dB, 0.001 deg - on every entry point):

  "fp32" : fp32 activations; every matrix operand is split into bf16 hi + lo and hi*hi + hi*lo + lo*hi is
           accumulated in fp32 on the same MFMA pipeline (~1e-5 relative; the fp32 parity gate);
  "fp16" : the package default, a POLICY rather than one kernel set: each UNet evaluation runs on the cheapest kernel set that
           keeps the caller's output inside the tolerance, decided by the GAIN with which the evaluation's error reaches that output:
             - the fp16 kernel set (fp16 activations and MFMA operands - 11-bit significands, the same kernels and rate as bf16,
               stores saturate at +-65504; the convolutions of the two high-resolution levels, Cout <= 128, multiply by fp16
               hi + lo WEIGHTS: the weight rounding is the one error of a 16-bit pipeline that is the same in every step of a chain,
               i.e. a bias, not noise - DESIGN.md section 5, tests/precision_emul.py) where the gain is below WIDE_STEP_GAIN;
             - the "fp32" kernel set otherwise: the steps of a reverse chain whose update feeds the denoiser's error into the
               state with a gain >= WIDE_STEP_GAIN (step_precision), and every BARE module call - UNet.forward, Block.forward,
               ResnetBlock.forward ... outside a sampler: the output is the result, gain 1 (forward_precision).  One forward of
               the UNet on the fp16 kernel set measures 1.07e-3 ... 1.16e-3 against the reference, i.e. outside the tolerance;
               the reverse chain multiplies it by 0.45 or less wherever that set runs.

EXPERIMENTAL kernel sets (A/B forms; outside the tolerance or redundant; refused unless allow_experimental() / HSIDM_EXPERIMENTAL=1):

  "bf16"   : bf16 activations and MFMA operands (8-bit significands): 7.9e-3 on the reference chains, 8x outside the tolerance;
             kept as the fastest form of the training step and as a throughput reference (BASELINE configs[1] names bf16);
  "fp16x1" / "fp16x2" : the fp16 policy with the second weight pass nowhere / wherever a kernel takes it.

The gain of a reverse step is |c_x0 * sqrt(1/acp_t - 1)| - the coefficient of eps in x_{t-1} (reference diffusion.py:142-149:
posterior_mean_coef1 * sqrt_recipm1_alphas_cumprod).  With the reference's cosine schedule it is 31.6 at the first step of EVERY
chain (beta is clamped at 0.999 there, diffusion.py:46) and 1.50 / 0.83 / 0.58 / 0.45 ... at the following steps whatever the chain
length (near t = T the schedule's alpha-bar is ~ (T - t)^2): four steps per chain - four in the shipped 20-step validation chain,
four in the benchmark's thousand.  (tests/precision_emul.py, key `is`: on the 20-step chain the first step alone carries 18 % of
the fp16 kernel set's deviation from the reference, the first two 21 %, the first six 39 %.)
"""
import contextlib
import os

# Package default (modules built without precision=...): the fastest mode that stays within the reference's tolerance
# (1e-3 / 0.01 dB / 0.001 deg on its validation chain AND on a bare forward); HSIDM_PRECISION or set_default_precision() choose another.
_default = os.environ.get("HSIDM_PRECISION", "fp16")


MODES = ("fp16", "fp32")
EXPERIMENTAL_MODES = ("bf16", "fp16x1", "fp16x2")
_experimental = bool(os.environ.get("HSIDM_EXPERIMENTAL"))
_as_named = False


def allow_experimental(flag=True):
    """Accept the EXPERIMENTAL kernel sets by name (they do not meet the reference's tolerance: measurement and A/B use)."""
    global _experimental
    old, _experimental = _experimental, bool(flag)
    return old


def _known(p):
    if p in MODES:
        return True
    if p in EXPERIMENTAL_MODES:
        if not _experimental:
            raise ValueError("hsidm: precision %r is an experimental kernel set outside the reference's tolerance (bf16: 7.9e-3 on the "
                             "reference chains); the public modes are %s - call hsi_dmgasr_amd.precision.allow_experimental() (or set "
                             "HSIDM_EXPERIMENTAL=1) to measure with it" % (p, (MODES,)))
        return True
    return False


def set_default_precision(p):
    global _default
    if not _known(p):
        raise ValueError("precision must be one of %s" % (MODES,))
    _default = p


def get_default_precision():
    return _default


def resolve_precision(p):
    p = _default if p is None else p
    if not _known(p):
        raise ValueError("precision must be one of %s, got %r" % (MODES, p))
    return p


WIDE_STEP_GAIN = 0.5


def step_precision(p, eps_gain):
    """Mode of ONE reverse step of a chain run in mode p, given the step's gain on the denoiser's output (see the module docstring)."""
    if p in ("fp16", "fp16x1", "fp16x2") and eps_gain >= WIDE_STEP_GAIN and not os.environ.get("HSIDM_NO_STEP_SCHEDULE"):
        return "fp32"
    return p


def forward_precision(p):
    """Kernel set of a BARE module call (UNet.forward, Block.forward, ... not issued by a sampler) in mode p: the call's output is
    the caller's result - gain 1 - so the "fp16" policy runs it on the fp32 kernel set (reference unet.py:239-263: what
    `netG.denoise_fn(x, t)` returns is held to 1e-3 like every other output of the path)."""
    return "fp32" if (p == "fp16" and not _as_named) else p


@contextlib.contextmanager
def kernels_as_named():
    """Inside: a bare module call in mode "fp16" runs the fp16 KERNEL SET itself, without the gain-1 widening (kernel tests, A/B
    measurements of single launches)."""
    global _as_named
    old, _as_named = _as_named, True
    try:
        yield
    finally:
        _as_named = old


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
