#!/usr/bin/env python3
"""Group-autoencoder encode / decode latency on the GPU (BASELINE configs[0] and [2] shapes) and the whole
per-image path (encode -> T-step sampler over all groups -> decode) for one CAVE patch.

    python tools/gae_bench.py [--steps T]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsi_dmgasr_amd import gae, pipeline  # noqa: E402
from hsi_dmgasr_amd.init import init_weights_orthogonal  # noqa: E402
from hsi_dmgasr_amd.sr3_modules import diffusion, unet  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--cubes", type=int, default=0, help="only this: encode / decode of N CAVE cubes 31x128x128 (bench.py's `gae` object at its "
                                                         "batch) a few times - the run to put under rocprofv3 for a per-kernel table / FETCH-WRITE passes")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    g = torch.Generator().manual_seed(3)
    if args.cubes:
        m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=8, n_ovls=2, n_colors=31, n_feats=64, precision=args.precision).to(dev).eval()
        x = torch.rand(args.cubes, 31, 128, 128, generator=g).to(dev)
        z = m.encode_batched(x)
        print(json.dumps(dict(cubes=args.cubes, precision=args.precision, encode_ms=timed(lambda: m.encode_batched(x), args.reps),
                              decode_ms=timed(lambda: m.decode_batched(z, 31), args.reps))))
        return
    for name, (ns, no, nc, hw, b) in {"cave_31x64x64": (8, 2, 31, 64, 1), "cave_31x128x128": (8, 2, 31, 128, 1),
                                      "cave_31x128x128_b8": (8, 2, 31, 128, 8),
                                      "chikusei_128x128x128": (16, 4, 128, 128, 1)}.items():
        for prec in ("fp32", "fp16"):
            m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=ns, n_ovls=no, n_colors=nc, n_feats=64, precision=prec).to(dev).eval()
            x = torch.rand(b, nc, hw, hw, generator=g).to(dev)
            z = m.encode_batched(x)
            out["%s/%s" % (name, prec)] = dict(encode_ms=timed(lambda: m.encode_batched(x)),
                                               decode_ms=timed(lambda: m.decode_batched(z, nc)), groups=m.G)
    # whole per-image path, one CAVE patch (5 latents), T steps
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                  attn_res=[16], res_blocks=2, dropout=0.2, image_size=128, precision="bf16")
    init_weights_orthogonal(u, seed=0)
    gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, conditional=True).to(dev).eval()
    gd.set_loss(dev)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=args.steps, linear_start=1e-6, linear_end=1e-2), dev)
    gd.noise = "philox"
    m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=8, n_ovls=2, n_colors=31, n_feats=64, precision="fp32").to(dev).eval()
    cube = torch.rand(1, 31, 128, 128, generator=g).to(dev)
    ms = timed(lambda: pipeline.super_resolve(m, gd, cube), reps=2)
    out["per_image_path"] = dict(T=args.steps, ms=ms, ms_per_step_batch5=ms / args.steps)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
