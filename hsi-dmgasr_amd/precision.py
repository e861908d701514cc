"""Precision mode of the HIP path.

PUBLIC modes.  Both are held to the reference's tolerance - 1e-3 relative on latents and cube, 0.01 dB, 0.001 deg - on every entry point.
"fp32" meets every bound by two orders of magnitude.  "fp16" meets every CONTINUOUS quantity on the eleven reference chains (tightest: 9.3e-4
on the latents the reference did not clamp); the reference's SAM INDEX (eval_hsi.py:47-65), which skips all-zero spectra and is therefore
discontinuous where a spectrum crosses the clamp at zero, is missed strictly on one hold-out chain (one pixel, 1.9e-3 deg; 3.9e-4 on the
common support): bench.py reports `meets_north_star` strictly (false) and tests/test_gpu_chain.py carries the miss as a named XFAIL.

  "fp32" : fp32 activations; every matrix operand is split into bf16 hi + lo and hi*hi + hi*lo + lo*hi is
           accumulated in fp32 on the same MFMA pipeline (~1e-5 relative; the fp32 parity gate);
  "fp16" : the package default, a POLICY rather than one kernel set: each UNet evaluation runs on the cheapest kernel set that
           keeps the caller's output inside the tolerance, decided by the GAIN with which the evaluation's error reaches that output:
             - the fp16 kernel sets "fp16d0" ... "fp16d3" (fp16 activations and MFMA operands - 11-bit significands, the same
               kernels and rate as bf16, stores saturate at +-65504 - ONE weight pass, the weights DITHERED over the steps of the
               chain: step k multiplies by fp16(w + d[k % 4] ulp(w)).  The weight rounding is the one error of a 16-bit pipeline
               that is the same in every step of a chain, i.e. a bias, not noise - DESIGN.md section 5, tests/precision_emul.py; the
               dither turns it into noise that averages out over four steps, without the second weight pass rounds 3-4 paid for it)
               where the gain is below WIDE_STEP_GAIN;
             - the "fp32h" kernel set (fp32 activations and GroupNorm pairs; every convolution rounds its staged operand ONCE to fp16
               and multiplies by fp16 hi + lo weights: two MFMA passes instead of three, include/hsidm.h HSIDM_F32H) on the steps whose
               gain lies in [WIDE_STEP_GAIN, FULL_STEP_GAIN): steps 2 .. 8 of a chain on the reference's cosine schedule (gains
               1.50 ... 0.27).  What has to go at those gains is the 11-bit STORAGE of the activations and the rounding of the
               weights - not the operand's 11 bits (emulated before it was built);
             - the "fp32" kernel set otherwise: the steps with a gain >= FULL_STEP_GAIN (the first step of every chain, gain 31.6), and
               every BARE module call - UNet.forward, Block.forward, ResnetBlock.forward ... outside a sampler: the output is the
               result, gain 1 (forward_precision).  One forward of the UNet on an fp16 kernel set measures 1.1e-3 ... 1.4e-3 against
               the reference, i.e. outside the tolerance; the reverse chain multiplies it by 0.24 or less wherever such a set runs.

EXPERIMENTAL kernel sets (A/B forms; outside the tolerance or redundant; refused unless allow_experimental() / HSIDM_EXPERIMENTAL=1):

  "bf16"   : bf16 activations and MFMA operands (8-bit significands): 7.9e-3 on the reference chains, 8x outside the tolerance;
             kept as the fastest form of the training step and as a throughput reference (BASELINE configs[1] names bf16);
  "fp16x1" / "fp16x2" : fp16 with plainly rounded weights in one pass / with hi + lo weights (the second pass 2:4-sparse on
             v_smfmac where a kernel takes it) wherever a kernel takes them; inside kernels_as_named() the name "fp16" is round 4's
             set, hi + lo weights on the Cout <= 128 layers.

The gain of a reverse step is |c_x0 * sqrt(1/acp_t - 1)| - the coefficient of eps in x_{t-1} (reference diffusion.py:142-149:
posterior_mean_coef1 * sqrt_recipm1_alphas_cumprod).  With the reference's cosine schedule it is 31.6 at the first step of EVERY
chain (beta is clamped at 0.999 there, diffusion.py:46) and 1.50 / 0.83 / 0.58 / 0.45 ... at the following steps whatever the chain
length (near t = T the schedule's alpha-bar is ~ (T - t)^2).
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


_internal = False            # inside the sampler's per-step dispatch (internal_names()): the dithered kernel-set names are accepted


def _known(p):
    if p in MODES:
        return True
    if dither_phase(p) is not None or p == "fp32h":
        # "fp16d<k>" / "fp32h" name ONE kernel set of the "fp16" policy.  Only the sampler's per-step choice (step_precision) produces it:
        # a module or a run built on it by name would skip the policy's fp32-set steps and the gain-1 widening of bare forwards -
        # a single forward on such a set measures 1.1e-3 ... 1.4e-3 - so by name it is an experimental set like the others
        if _internal or _experimental:
            return True
        raise ValueError("hsidm: precision %r is one kernel set of the \"fp16\" policy (chosen per chain step by the sampler), not a mode; "
                         "use \"fp16\", or allow_experimental() to run the set by name" % (p,))
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


# Steps whose update passes a quarter or more of the denoiser's output error on to the state run on the fp32 kernel set: with the
# reference's cosine schedule the gains of a chain's steps are 31.6, 1.50, 0.83, 0.58, 0.45, 0.37, 0.31, 0.27 | 0.24, 0.21 ... whatever
# its length (~ 1.5 / k behind the first), i.e. EIGHT steps per chain (0.29 ms per step averaged over the metric's 1000-step chain).
# The chain's deviation is made where the gain is large; emulated on the T = 20 reference chains (tests/precision_emul.py, one-pass
# dithered weights, synth:0 / orth:1): 4 steps 6.8e-4 / 7.7e-4, 6 steps 5.4e-4 / 6.1e-4, 8 steps 4.9e-4 / 5.3e-4 (dSAM 4e-6 / 2.7e-4 deg);
# hi + 2:4-sparse lo weights with 4 steps (round 4's policy) 5.5e-4 / 5.7e-4.  Each further step costs 0.16 % of a 1000-step chain.
# (HSIDM_WIDE_STEP_GAIN: the threshold is a knob between accuracy and speed on SHORT chains - 1/6 makes it twelve steps: measured
# 2.8e-4 ... 3.0e-4 instead of 4.9e-4 ... 5.5e-4 on the 20-step reference chains, nothing on the 1000-step ones (3.0e-4 either way) nor on
# the SAM index, for -0.5 % on the 1000-step benchmark (10 217 against 10 268 in one box) and +19 % time on a 20-step chain)
WIDE_STEP_GAIN = float(os.environ.get("HSIDM_WIDE_STEP_GAIN", "0.25"))
# ... and of those, the steps below FULL_STEP_GAIN run on the "fp32h" set (two MFMA passes) instead of "fp32" (three): with the default
# only the first step of a chain (gain 31.6; the next is 1.50) keeps the full set.  HSIDM_FULL_STEP_GAIN=0 (or HSIDM_NO_FP32H=1): every
# wide step on "fp32" (round 5's policy, A/B).
FULL_STEP_GAIN = 0.0 if os.environ.get("HSIDM_NO_FP32H") else float(os.environ.get("HSIDM_FULL_STEP_GAIN", "2.0"))

# Weight dither of the "fp16" policy's chain steps.  The fp16 rounding of a WEIGHT is the one error of a 16-bit pipeline that repeats
# in every step of a chain - a bias, which is what moves the quality indices.  The fp16 kernel set of chain
# step k therefore multiplies by  fp16(w + d[k % K] * ulp(w))  with DITHER_K offsets d spread over (-1/2, 1/2) (bit-reversed order):
# over K consecutive steps every weight is rounded up in about frac * K of them (frac: its position in its rounding interval), so
# the MEAN weight the chain sees is w to 1 / (2 K) ulp.  Kernel-set names: "fp16d0" ... "fp16d<K-1>" (produced by step_precision
# only; one packed copy of the 16-bit weights and one captured graph per phase).  HSIDM_DITHER_K=0: off (A/B: the plainly rounded one-pass
# set "fp16x1" on the fp16-set steps).
DITHER_K = int(os.environ.get("HSIDM_DITHER_K", "4"))


def dither_phase(p):
    """(phase, K) of a dithered kernel-set name "fp16d<phase>", else None."""
    if isinstance(p, str) and p.startswith("fp16d") and p[5:].isdigit() and int(p[5:]) < DITHER_K:
        return int(p[5:]), DITHER_K
    return None


def dither_offset(phase, K):
    """Offset (in ulp of the weight) of phase `phase` of K: ((bit-reversed phase) + 1/2) / K - 1/2."""
    if K & (K - 1) == 0:
        bits = max(1, (K - 1).bit_length())
        phase = int(format(phase % K, "0%db" % bits)[::-1], 2)
    return (phase % K + 0.5) / K - 0.5


def family(p):
    """Kernel-set family of a name: the dither phases are one set ("fp16d2" -> "fp16"); "fp32" and "fp32h" are their own."""
    return "fp16" if dither_phase(p) is not None else p


def step_precision(p, eps_gain, step=0):
    """Kernel set of ONE reverse step (the step-th of its chain) of a chain run in mode p, given the step's gain on the denoiser's
    output (see the module docstring)."""
    if p in ("fp16", "fp16x1", "fp16x2") and eps_gain >= WIDE_STEP_GAIN and not os.environ.get("HSIDM_NO_STEP_SCHEDULE"):
        return "fp32h" if eps_gain < FULL_STEP_GAIN else "fp32"
    if p == "fp16":
        # (HSIDM_DITHER_K <= 1: the A/B form without the dither is the plainly rounded one-pass set, "fp16x1" - NOT the name "fp16",
        # which as a kernel-set name is round 4's hi + lo set on Cout <= 128)
        return "fp16d%d" % (step % DITHER_K) if DITHER_K > 1 else "fp16x1"
    return p


def forward_precision(p):
    """Kernel set of a BARE module call (UNet.forward, Block.forward, ... not issued by a sampler) in mode p: the call's output is
    the caller's result - gain 1 - so the "fp16" policy runs it on the fp32 kernel set (reference unet.py:239-263: what
    `netG.denoise_fn(x, t)` returns is held to 1e-3 like every other output of the path)."""
    return "fp32" if (p == "fp16" and not _as_named) else p


@contextlib.contextmanager
def internal_names():
    """Inside: the kernel-set names step_precision produces ("fp16d<k>") resolve (the sampler's per-step dispatch)."""
    global _internal
    old, _internal = _internal, True
    try:
        yield
    finally:
        _internal = old


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
    return p not in ("fp32", "fp32h")


def wide_weights(p, cout, cin=0, ksize=3):
    """Does a convolution (cout x cin x ksize x ksize) carry hi + lo weights in kernel set p (second MFMA pass)?  Only the experimental
    two-pass sets: "fp16x2" everywhere a kernel takes them, and the NAME "fp16" taken as a kernel set (kernels_as_named(): round 4's
    set) on the Cout <= 128 layers; the policy's dithered sets "fp16d<k>", "fp16x1" and bf16 are one-pass."""
    return p == "fp16x2" or (p == "fp16" and cout <= 128)
