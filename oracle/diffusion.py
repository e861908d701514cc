"""Oracle: SR3 Gaussian-diffusion schedule and sampler (fp32 CPU, float64 schedule).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates model/sr3_modules/diffusion.py of the reference.  The network is passed
in as ``denoise(x_cat, gamma) -> eps`` so the same loop can drive the oracle
UNet or any other callable.
"""
import math

import numpy as np
import torch

from . import philox


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """diffusion.py:19-49, all schedules, float64 numpy result."""
    T = n_timestep

    def warm(frac):
        b = linear_end * np.ones(T, dtype=np.float64)
        k = int(T * frac)
        b[:k] = np.linspace(linear_start, linear_end, k, dtype=np.float64)
        return b

    if schedule == "quad":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, T, dtype=np.float64) ** 2
    if schedule == "linear":
        return np.linspace(linear_start, linear_end, T, dtype=np.float64)
    if schedule == "warmup10":
        return warm(0.1)
    if schedule == "warmup50":
        return warm(0.5)
    if schedule == "const":
        return linear_end * np.ones(T, dtype=np.float64)
    if schedule == "jsd":
        return 1.0 / np.linspace(T, 1, T, dtype=np.float64)
    if schedule == "cosine":
        # the reference evaluates this branch in torch float64 (diffusion.py:37-46)
        ts = torch.arange(T + 1, dtype=torch.float64) / T + cosine_s
        ac = torch.cos(ts / (1 + cosine_s) * math.pi / 2).pow(2)
        ac = ac / ac[0]
        betas = (1 - ac[1:] / ac[:-1]).clamp(max=0.999)
        return betas.numpy()
    raise NotImplementedError(schedule)


BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def noise_schedule(schedule_opt):
    """set_new_noise_schedule diffusion.py:93-140.

    Returns dict with the 12 fp32 buffers (numpy float32, length T) and the
    float64 ``sqrt_alphas_cumprod_prev`` (length T+1, *not* a buffer, :106-107).
    """
    betas = make_beta_schedule(schedule_opt["schedule"], schedule_opt["n_timestep"],
                               schedule_opt["linear_start"], schedule_opt["linear_end"])
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    f64 = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.maximum(post_var, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }
    out = {k: v.astype(np.float32) for k, v in f64.items()}
    out["sqrt_alphas_cumprod_prev"] = np.sqrt(np.append(1.0, ac))
    out["num_timesteps"] = int(betas.shape[0])
    return out


def p_sample_step(denoise, sched, x, cond, i, z):
    """One reverse step: p_mean_variance + p_sample, diffusion.py:152-175.

    x, cond (B,3,H,W) fp32; i = python int loop index; z = N(0,1) noise (ignored at i == 0).
    """
    b = x.shape[0]
    gamma = torch.full((b, 1), float(np.float32(sched["sqrt_alphas_cumprod_prev"][i + 1])),
                       dtype=torch.float32)
    inp = torch.cat([cond, x], dim=1) if cond is not None else x
    eps = denoise(inp, gamma)
    f = lambda name: torch.tensor(sched[name][i], dtype=torch.float32)
    x0 = f("sqrt_recip_alphas_cumprod") * x - f("sqrt_recipm1_alphas_cumprod") * eps
    x0 = x0.clamp(-1.0, 1.0)
    mean = f("posterior_mean_coef1") * x0 + f("posterior_mean_coef2") * x
    if i > 0:
        return mean + z * (0.5 * f("posterior_log_variance_clipped")).exp()
    return mean


def p_sample_loop(denoise, sched, cond, x_T, noise_fn, continous=False):
    """p_sample_loop diffusion.py:177-201, conditional branch.

    noise_fn(i) -> z for loop index i (i = T-1 .. 1).  Returns ret_img (continous)
    or ret_img[-1] (the last *sample of the batch*, reference quirk F8).
    """
    T = sched["num_timesteps"]
    inter = 1 | (T // 10)
    img = x_T
    ret = cond
    for i in reversed(range(T)):
        z = noise_fn(i) if i > 0 else None
        img = p_sample_step(denoise, sched, img, cond, i, z)
        if i % inter == 0:
            ret = torch.cat([ret, img], dim=0)
    return ret if continous else ret[-1]


def p_sample_loop_unconditional(denoise, sched, x_T, noise_fn, continous=False):
    """p_sample_loop diffusion.py:177-189, the `not self.conditional` branch (what sample() :203-207 runs): the network sees
    x_t alone and ret_img starts with x_T itself."""
    T = sched["num_timesteps"]
    inter = 1 | (T // 10)
    img = x_T
    ret = img
    for i in reversed(range(T)):
        img = p_sample_step(denoise, sched, img, None, i, noise_fn(i) if i > 0 else None)
        if i % inter == 0:
            ret = torch.cat([ret, img], dim=0)
    return ret if continous else ret[-1]


def philox_noise_fn(seed, shape):
    """Noise source mirroring the device sampler's counter-based generator:
    stream index = loop index i for the per-step noise, T for x_T."""
    n = int(np.prod(shape))

    def fn(i):
        return torch.from_numpy(philox.normal(seed, i, n).reshape(shape))
    return fn


def q_sample(x0, gamma, noise):
    """diffusion.py:213-220."""
    return gamma * x0 + (1 - gamma ** 2).sqrt() * noise


def p_losses(denoise, x_hr, x_sr, gamma, noise, loss_type="l1"):
    """p_losses diffusion.py:222-250 with injected (gamma, noise); sum-reduced loss."""
    x_noisy = q_sample(x_hr, gamma.view(-1, 1, 1, 1), noise)
    rec = denoise(torch.cat([x_sr, x_noisy], dim=1), gamma.view(-1, 1))
    if loss_type == "l1":
        return (noise - rec).abs().sum()
    if loss_type == "l2":
        return ((noise - rec) ** 2).sum()
    raise NotImplementedError(loss_type)


# ---------------------------------------------------------------------------------------------------
# Strided DDIM sampler.  NOT in the reference (it only has the ancestral sampler above):
# restated from Song, Meng & Ermon, "Denoising Diffusion Implicit Models" (ICLR 2021), eq. 12 / 16, on the
# reference's schedule buffers.  Pin: with eta = 1 and all T steps its tables equal the reference's
# posterior tables (tests/test_oracle_golden.py::test_ddim_tables_reduce_to_the_reference_posterior).
# ---------------------------------------------------------------------------------------------------
def ddim_schedule(schedule_opt, steps, eta=0.0):
    """Tables of a K-step sampler on the sub-sequence tau = round(linspace(0, T-1, K)) of the reference schedule.

    Row i (loop index, i = K-1 .. 0) holds what p_sample_step takes from the reference buffers at index t:
    x0 = sqrt_recip*x - sqrt_recipm1*eps (clamped, diffusion.py:171), x_prev = coef_x0*x0 + coef_xt*x + sigma*z, where
    eps is re-derived from the clamped x0:  coef_xt = c/sqrt(1-a_t), coef_x0 = sqrt(a_prev) - coef_xt*sqrt(a_t),
    c = sqrt(1 - a_prev - sigma^2), sigma = eta*sqrt((1-a_prev)/(1-a_t))*sqrt(1 - a_t/a_prev).  ``level[i+1]`` is the
    noise level handed to the UNet (sqrt(a_tau_i), the reference's sqrt_alphas_cumprod_prev[t+1]).
    """
    betas = np.asarray(make_beta_schedule(schedule_opt["schedule"], schedule_opt["n_timestep"],
                                          schedule_opt["linear_start"], schedule_opt["linear_end"]), dtype=np.float64)
    T = betas.shape[0]
    K = int(steps)
    assert 1 <= K <= T
    ac = np.cumprod(1.0 - betas)
    tau = np.round(np.linspace(0, T - 1, K)).astype(np.int64) if K > 1 else np.array([T - 1], dtype=np.int64)
    a_t = ac[tau]
    a_prev = np.append(1.0, a_t[:-1])
    sigma2 = (eta ** 2) * (1.0 - a_prev) / (1.0 - a_t) * (1.0 - a_t / a_prev)
    c = np.sqrt(np.maximum(1.0 - a_prev - sigma2, 0.0))
    coef_xt = c / np.sqrt(1.0 - a_t)
    return {
        "tau": tau, "num_timesteps": K,
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / a_t).astype(np.float32),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / a_t - 1).astype(np.float32),
        "coef_x0": (np.sqrt(a_prev) - coef_xt * np.sqrt(a_t)).astype(np.float32),
        "coef_xt": coef_xt.astype(np.float32),
        "log_sigma2": np.log(np.maximum(sigma2, 1e-20)).astype(np.float32),
        "level": np.sqrt(np.append(1.0, a_t)),
    }


def ddim_step(denoise, tab, x, cond, i, z):
    """One step of the strided sampler; same shape as p_sample_step (z ignored at i == 0)."""
    b = x.shape[0]
    gamma = torch.full((b, 1), float(np.float32(tab["level"][i + 1])), dtype=torch.float32)
    inp = torch.cat([cond, x], dim=1) if cond is not None else x
    eps = denoise(inp, gamma)
    f = lambda name: torch.tensor(tab[name][i], dtype=torch.float32)
    x0 = (f("sqrt_recip_alphas_cumprod") * x - f("sqrt_recipm1_alphas_cumprod") * eps).clamp(-1.0, 1.0)
    mean = f("coef_x0") * x0 + f("coef_xt") * x
    if i > 0:
        return mean + z * (0.5 * f("log_sigma2")).exp()
    return mean


def ddim_sample_loop(denoise, tab, cond, x_T, noise_fn):
    """K-step reverse process; returns the final x (all samples of the batch)."""
    img = x_T
    for i in reversed(range(tab["num_timesteps"])):
        img = ddim_step(denoise, tab, img, cond, i, noise_fn(i) if i > 0 else None)
    return img
