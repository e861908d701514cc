#!/usr/bin/env python3
"""Full-size, full-length chain golden (chain.npz) by IMPORTING THE REFERENCE (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_chain.py

The reference's own validation iteration (sr_gae.py:436-474) at its shipped configuration: the 97.8 M-parameter UNet
(config/sr_sr3_16_128.json:78-94), its T = 20 cosine chain (:96-107), one CAVE image = 5 spectral groups of the
pretrained CAVE group-autoencoder (GAE_pretrained/GAE_4_Cav.pth), 31 x 128 x 128:

    z = GAE.encode(SR); per group: x_0 = GaussianDiffusion.super_resolution(z_g) (diffusion.py:177-211); y = GAE.decode(x_0)

torch.randn / randn_like inside the sampler are replaced by tests/golden/synth.py:chain_noise so that the tests can
regenerate the very same draws.  Stored: the five denoised latents, the decoded cube, and the reference's own quality
indices of that cube against the synthetic ground truth.  UNet weights: synth_param("unet_full." + key).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, T, _import_reference, fill  # noqa: E402
from synth import CHAIN_T, chain_cubes, chain_noise  # noqa: E402


def main():
    unet, diff, AE = _import_reference()
    import eval_hsi
    torch.set_num_threads(8)
    hr, sr = chain_cubes()
    g = torch.load(REF + "/GAE_pretrained/GAE_4_Cav.pth", map_location="cpu", weights_only=False).eval()
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                  attn_res=[16], res_blocks=2, dropout=0.2, image_size=128)
    fill(u, "unet_full.")
    opt = dict(schedule="cosine", n_timestep=CHAIN_T, linear_start=1e-6, linear_end=1e-2)
    gd = diff.GaussianDiffusion(u, image_size=128, channels=3, conditional=True)
    gd.set_loss("cpu")
    gd.set_new_noise_schedule(opt, "cpu")
    gd.eval()

    real_randn, real_like = torch.randn, torch.randn_like
    out = {}
    with torch.no_grad():
        x = T(sr)
        zs = [g.Encoder(x[:, s:e]) for s, e in zip(g.start_idx, g.end_idx)]          # sr_gae.py:456 -> AE.py:310-324
        x0s = []
        for gi, z in enumerate(zs):                                                   # sr_gae.py:458-465
            draws = iter(range(CHAIN_T))
            torch.randn = lambda *a, **k: T(chain_noise(gi, next(draws)))
            torch.randn_like = lambda t, **k: T(chain_noise(gi, next(draws)))
            try:
                x0s.append(gd.super_resolution(z, continous=False).unsqueeze(0))
            finally:
                torch.randn, torch.randn_like = real_randn, real_like
            assert next(draws, None) is None, "the sampler drew fewer tensors than expected"
            print("group %d done" % gi, flush=True)
        y = torch.zeros_like(x)
        cnt = torch.zeros(x.shape[1])
        for (s, e), z in zip(zip(g.start_idx, g.end_idx), x0s):                       # sr_gae.py:467 -> AE.py:283-308
            y[:, s:e] += g.Decoder(z)
            cnt[s:e] += 1
        y = y / cnt[None, :, None, None]
        y = (g.final(g.trunk(y)) + y).clamp(0, 1)                                     # sr_gae.py:474
    out["z"] = torch.cat(zs).numpy()
    out["x0"] = torch.cat(x0s).numpy()
    out["y"] = y.numpy()
    a = hr[0].transpose(1, 2, 0)
    b = out["y"][0].transpose(1, 2, 0)
    out["sam"] = np.array(eval_hsi.compare_sam(a, b))                                 # eval_hsi.py:47-65 (degrees)
    out["rmse"] = np.array(eval_hsi.compare_rmse(a, b))
    # skimage is absent here: MPSNR from its documented formula in float64 (as metrics.npz)
    out["mpsnr_formula"] = np.array(np.mean([10 * np.log10(1.0 / np.mean((a[:, :, k].astype(np.float64) - b[:, :, k]) ** 2))
                                             for k in range(a.shape[2])]))
    np.savez_compressed(os.path.join(HERE, "chain.npz"), **out)
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})
    print("chain.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "chain.npz")) / 1024))


if __name__ == "__main__":
    main()
