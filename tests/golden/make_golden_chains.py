#!/usr/bin/env python3
"""The chain fixture SET (chains/<weights>_n<draw>_T<steps>.npz) by IMPORTING THE REFERENCE (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_chains.py orth:0:20 orth:1:20 synth:1:20 orth:2:1000 synth:3:1000 chi:orth:3:20

Same run as make_golden_chain.py (chain.npz = "synth:0:20", kept as it is) - the reference's validation iteration,
sr_gae.py:436-474, at the shipped configuration on one CAVE image - over

  * two weight sets: "synth" = synth_param("unet_full." + key) as in chain.npz; "orth" = the reference's OWN initialisation,
    model/networks.py:45-57,110-112 (`init_weights(netG, init_type='orthogonal')`) under torch.manual_seed(0) - the weights
    bench.py times.  The tests rebuild them with torch's nn.init.orthogonal_ under the same seed; the fixture carries three
    numbers per tensor (norm and two fixed random projections) so that the rebuild is CHECKED, not assumed;
  * several noise draws: draw d uses synth_tensor("chain.noise.g%d.k%d", seed=d) (draw 0 = chain_noise of chain.npz);
  * the chain length T: 20 (the shipped validation setting, config/sr_sr3_16_128.json:96-107) or 1000 (BASELINE.json's
    metric: the 1000-step p_sample_loop, diffusion.py:177-201), cosine schedule.

  * the data set: "cave" (31 bands, 5 groups; GAE_4_Cav.pth) or "chi" = Chikusei, BASELINE configs[2] (128 bands, n_subs 16 / n_ovls 4 ->
    11 groups; the pretrained GAE_4_Chi.pth, whose tensors are stored as gae_chi_state.npz): chains/chi_<weights>_n<draw>_T<steps>.npz
    holds the eleven denoised latents, every fourth band of the decoded cube and the reference's indices of the whole cube.

The conditioning cube of draw d is chain_cubes() for d = 0 and its d-th variant otherwise (synth.chain_cubes_draw).
Stored: z (the reference encoder's latents), x0 (five denoised latents), y (decoded cube), the reference's indices.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, T, _import_reference, fill  # noqa: E402
from synth import chain_cubes_draw, chain_noise_draw, weight_probe  # noqa: E402


def _load_networks():
    import importlib.util
    s = importlib.util.spec_from_file_location("ref_networks", REF + "/model/networks.py")
    m = importlib.util.module_from_spec(s)
    s.loader.exec_module(m)
    return m


DATASETS = {"cave": ("GAE_4_Cav.pth", 31), "chi": ("GAE_4_Chi.pth", 128)}     # BASELINE configs[1] / configs[2] (Chikusei: 128 bands, G = 11)


def make(weights, draw, steps, threads, dataset="cave"):
    unet, diff, AE = _import_reference()
    import eval_hsi
    torch.set_num_threads(threads)
    ckpt, bands = DATASETS[dataset]
    hr, sr = chain_cubes_draw(draw, bands)
    g = torch.load(REF + "/GAE_pretrained/" + ckpt, map_location="cpu", weights_only=False).eval()
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                  attn_res=[16], res_blocks=2, dropout=0.2, image_size=128)
    gd = diff.GaussianDiffusion(u, image_size=128, channels=3, conditional=True)
    out = {}
    if weights == "synth":
        fill(u, "unet_full.")
    else:
        torch.manual_seed(0)
        _load_networks().init_weights(gd, init_type="orthogonal")                       # model/networks.py:110-112
        keys, probes = [], []
        for k, v in u.state_dict().items():
            keys.append(k)
            probes.append(weight_probe(k, v.numpy()))
        out["w_keys"] = np.array(keys)
        out["w_probe"] = np.array(probes, dtype=np.float64)
    opt = dict(schedule="cosine", n_timestep=steps, linear_start=1e-6, linear_end=1e-2)
    gd.set_loss("cpu")
    gd.set_new_noise_schedule(opt, "cpu")
    gd.eval()

    real_randn, real_like = torch.randn, torch.randn_like
    t0 = time.time()
    with torch.no_grad():
        x = T(sr)
        zs = [g.Encoder(x[:, s:e]) for s, e in zip(g.start_idx, g.end_idx)]          # sr_gae.py:456 -> AE.py:310-324
        x0s = []
        for gi, z in enumerate(zs):                                                   # sr_gae.py:458-465
            draws = iter(range(steps))
            torch.randn = lambda *a, **k: T(chain_noise_draw(draw, gi, next(draws)))
            torch.randn_like = lambda t, **k: T(chain_noise_draw(draw, gi, next(draws)))
            try:
                x0s.append(gd.super_resolution(z, continous=False).unsqueeze(0))
            finally:
                torch.randn, torch.randn_like = real_randn, real_like
            assert next(draws, None) is None, "the sampler drew fewer tensors than expected"
            print("%s:%d:%d group %d done (%.0f s)" % (weights, draw, steps, gi, time.time() - t0), flush=True)
        y = torch.zeros_like(x)
        cnt = torch.zeros(x.shape[1])
        for (s, e), z in zip(zip(g.start_idx, g.end_idx), x0s):                       # sr_gae.py:467 -> AE.py:283-308
            y[:, s:e] += g.Decoder(z)
            cnt[s:e] += 1
        y = y / cnt[None, :, None, None]
        y = (g.final(g.trunk(y)) + y).clamp(0, 1)                                     # sr_gae.py:474
    out["x0"] = torch.cat(x0s).numpy()
    if dataset == "cave":
        out["z"] = torch.cat(zs).numpy()
        out["y"] = y.numpy()
    else:       # 128 bands: every fourth band of the cube (2 MB instead of 8), the indices below are the reference's on the WHOLE cube
        out["y_sub4"] = y.numpy()[:, ::4]
        st = os.path.join(HERE, "gae_%s_state.npz" % dataset)
        if not os.path.exists(st):      # the pretrained autoencoder of this data set, as data (like gae_cav_state.npz)
            np.savez_compressed(st, **{k: v.numpy() for k, v in g.state_dict().items()})
        out["groups"] = np.array([g.G, len(g.start_idx)])
    a = hr[0].transpose(1, 2, 0)
    b = y.numpy()[0].transpose(1, 2, 0)
    out["sam"] = np.array(eval_hsi.compare_sam(a, b))                                 # eval_hsi.py:47-65 (degrees)
    out["rmse"] = np.array(eval_hsi.compare_rmse(a, b))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import metrics as ometrics           # the test-side restatement on the SAME cube: what a test that only has part of y compares with
    out["sam_oracle"] = np.array(ometrics.sam_degrees(a, b))
    out["mpsnr_formula"] = np.array(np.mean([10 * np.log10(1.0 / np.mean((a[:, :, k].astype(np.float64) - b[:, :, k]) ** 2))
                                             for k in range(a.shape[2])]))
    os.makedirs(os.path.join(HERE, "chains"), exist_ok=True)
    path = os.path.join(HERE, "chains", "%s%s_n%d_T%d.npz" % ("" if dataset == "cave" else dataset + "_", weights, draw, steps))
    np.savez_compressed(path, **out)
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if v.dtype.kind == "f"})
    print("%s %.1f KB, clamped latent fraction %.3f, zero cube fraction %.3f" % (
        path, os.path.getsize(path) / 1024, float((np.abs(out["x0"]) >= 1.0).mean()), float((y.numpy() == 0).mean())))


if __name__ == "__main__":
    threads = int(os.environ.get("CHAIN_THREADS", "8"))
    for spec in sys.argv[1:]:           # weights:draw:T  or  dataset:weights:draw:T
        f = spec.split(":")
        ds = f.pop(0) if len(f) == 4 else "cave"
        make(f[0], int(f[1]), int(f[2]), threads, ds)
