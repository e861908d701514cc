#!/usr/bin/env python3
"""Drift of the 16-bit modes (fp16 = the headline mode, bf16) over long chains: the shipped UNet, one CAVE image's 5 group latents, cosine schedule with T steps,
the same Philox noise in both modes; the fp32 mode (within 1e-5 of the reference over the reference's own 20-step chain,
tests/test_gpu_chain.py) is the yardstick.  Prints one JSON line per T:

    python tools/drift.py [--T 20 100 1000] [--modes fp16 bf16]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import chain_cubes, synth_param  # noqa: E402
from hsi_dmgasr_amd import gae, pipeline  # noqa: E402
from hsi_dmgasr_amd.sr3_modules import diffusion, unet  # noqa: E402
from oracle import metrics  # noqa: E402  (quality indices of the two decoded cubes: a measurement tool, not the product)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, nargs="+", default=[20, 100, 1000])
    ap.add_argument("--modes", nargs="+", default=["fp16", "bf16"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hr, sr = chain_cubes()
    m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=8, n_ovls=2, n_colors=31, n_feats=64, precision="fp32").to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "gae_cav_state.npz")).items()})
    out = {}
    for prec in ["fp32"] + list(args.modes):
        u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8], attn_res=[16],
                      res_blocks=2, dropout=0.2, image_size=128, precision=prec).to(dev).eval()
        u.load_state_dict({k: torch.from_numpy(synth_param("unet_full." + k, tuple(v.shape))) for k, v in u.state_dict().items()})
        gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, conditional=True)
        gd.set_loss(dev)
        for T in args.T:
            gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=T, linear_start=1e-6, linear_end=1e-2), dev)
            gd.noise, gd.seed = "philox", 12345
            y, lat = pipeline.super_resolve(m, gd, torch.from_numpy(sr).to(dev), precision=prec)
            out[(prec, T)] = (y.cpu().numpy(), lat.cpu().numpy())
        del u, gd
        torch.cuda.empty_cache()
    a = hr[0].transpose(1, 2, 0)
    for mode in args.modes:
        for T in args.T:
            y32, l32 = out[("fp32", T)]
            y16, l16 = out[(mode, T)]
            g32, g16 = y32[0].transpose(1, 2, 0), y16[0].transpose(1, 2, 0)
            print(json.dumps({"mode": mode, "T": T, "latent_rel_diff": float(np.linalg.norm(l16 - l32) / np.linalg.norm(l32)),
                              "cube_rel_diff": float(np.linalg.norm(y16 - y32) / np.linalg.norm(y32)),
                              "dPSNR_dB": abs(metrics.mpsnr(a, g16) - metrics.mpsnr(a, g32)),
                              "dSAM_deg": abs(metrics.sam_degrees(a, g16) - metrics.sam_degrees(a, g32)),
                              "psnr_vs_fp32_cube_dB": metrics.mpsnr(g32, g16)}), flush=True)

if __name__ == "__main__":
    main()
