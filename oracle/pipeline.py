"""Oracle: the per-image inference path of sr_gae.py:436-494 (fp32 CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

encode SR cube -> for each spectral group run the conditional sampler on its
(1,3,H,W) latent -> decode the G denoised latents -> clamp to [0,1].
The reference runs the G groups sequentially at batch 1 (sr_gae.py:458-465);
they are independent, so ``batched=True`` stacks them on the batch axis.
"""
import torch

from . import diffusion, gae, sr3_unet


def super_resolve_cube(unet_sd, unet_cfg, gae_sd, gae_cfg, sched, sr_cube, noise_for_group):
    """sr_cube (1,C,H,W) in [0,1]; noise_for_group(g) -> (x_T, noise_fn) for group g.

    Returns (decoded cube (C,H,W) clamped to [0,1], list of denoised latents).
    """
    n_subs, n_ovls = gae_cfg["n_subs"], gae_cfg["n_ovls"]
    z_sr = gae.gae_encode(gae_sd, sr_cube, n_subs, n_ovls)
    denoise = lambda x, gamma: sr3_unet.unet_forward(unet_sd, unet_cfg, x, gamma)
    out = []
    for g, z in enumerate(z_sr):
        x_T, nf = noise_for_group(g)
        x0 = diffusion.p_sample_loop(denoise, sched, z, x_T, nf, continous=False)   # (3,H,W)
        out.append(x0.unsqueeze(0))
    y = gae.gae_decode(gae_sd, sr_cube.shape[1], out, n_subs, n_ovls)
    return y.clamp(0.0, 1.0)[0], out
