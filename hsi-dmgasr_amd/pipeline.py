"""Per-image inference path: encode the low-resolution cube, denoise every spectral-group latent, decode.

Counterpart of the reference's validation loop (sr_gae.py:436-494): there the G groups of one image are
sampled by G sequential batch-1 p_sample_loop calls with a device->host->device round trip per group
(sr_gae.py:458-465); here all groups of all patches of this rank form one batch that stays on the device.
"""
import torch

from . import degrade, metrics, parallel


@torch.no_grad()
def super_resolve(gae, gd, sr_cubes, x_T=None, noise=None, precision=None):
    """sr_cubes [P, C, H, W] in [0,1] (bicubic-upsampled LR cubes) -> (SR cubes [P, C, H, W] clamped to [0,1],
    denoised latents [P, G, 3, H, W]).  x_T / noise (optional): [P*G,3,H,W] / [T-1, P*G,3,H,W] injected noise."""
    P, C, H, W = sr_cubes.shape
    z = gae.encode_batched(sr_cubes)                               # [P, G, 3, H, W]
    G = z.shape[1]
    cond = z.reshape(P * G, 3, H, W).contiguous()
    x0 = gd.p_sample_loop_batched(cond, x_T=x_T, noise=noise, precision=precision)
    lat = x0.view(P, G, 3, H, W)
    y = gae.decode_batched(lat, C)
    return y.clamp_(0.0, 1.0), lat


@torch.no_grad()
def super_resolve_sharded(gae, gd, sr_cubes_all, precision=None):
    """Patches sharded contiguously over the ranks of the default process group; every rank returns all cubes.
    (weights are expected to have been broadcast once with parallel.broadcast_module_)."""
    fn = lambda cubes: super_resolve(gae, gd, cubes, precision=precision)[0]
    return parallel.run_sharded(sr_cubes_all, fn)


@torch.no_grad()
def evaluate(gae, gd, raw_cubes, n_scale=4, normalize=True, precision=None):
    """The whole validation iteration of sr_gae.py:436-494 on the device, for a batch of ground-truth cubes
    raw_cubes [P, C, H, W]: min-max normalise (HStest.py:37) -> bicubic x1/n, xn (HStest.py:43-45) -> encode, denoise every
    group latent with the sampler set on `gd` (set_sampler), decode -> quality indices against the ground truth
    (eval_hsi.py).  Returns (SR cubes, indices [P, 6] with columns metrics.NAMES, bicubic baseline indices [P, 6])."""
    gt = degrade.minmax_normalize(raw_cubes) if normalize else raw_cubes.to(torch.float32).contiguous()
    _, lms = degrade.lr_pair(gt, n_scale)
    sr, _ = super_resolve(gae, gd, lms, precision=precision)
    return sr, metrics.quality_indices(gt, sr, ratio=n_scale), metrics.quality_indices(gt, lms, ratio=n_scale)
