#!/usr/bin/env python3
"""Benchmark of the denoising hot path: SR3 UNet p_sample steps over GAE-latent batches on MI355X.

    python bench.py --gpus N --steps K --warmup W            (+ --detail for the secondary legs)

N > 1 from a plain shell: bench.py starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process
(before anything touches the GPU) and exits with its return code; under torch.distributed.run (RANK / WORLD_SIZE set) it is
one rank per GPU over RCCL.  --total-patches P: strong scaling (BASELINE configs[3]: P patches sharded over the ranks).

A "step" is one reverse-diffusion step (noise embedding + full 97.8 M-parameter UNet forward + fused
posterior update) over this GPU's batch of latents: `--patches` CAVE patches x 5 spectral groups, each a
(3,128,128) latent conditioned on its low-resolution latent (BASELINE.json configs[1]; cosine T=1000).
Inputs are resident in HBM before the timed region; the step is a captured HIP graph.
value = steps * total batch / seconds  ("UNet denoise-steps/sec x batch").

OUTPUT.  The LAST line of stdout (rank 0) is ONE compact JSON object (< 4 KB, `compact_line`): the contract fields, `roofline`
(dominant kernel instance + the conv_v3 family as one row + the whole step), `cpu_baseline` (both cases) and `parity` (the
headline mode's worst case over the reference-chain fixtures).  Everything else - per-fixture parity, every kernel instance's
roofline row, and with --detail the other precision modes, small batches, the group autoencoder, the training step and the
1000-step reference chains - goes to the side file `bench_detail.json` (--detail-out), never to stdout.

The headline precision mode is "fp16" (precision.py: fp16 storage and MFMA operands, ONE weight pass with the weights dithered
over the chain's steps so that their rounding averages out instead of biasing the chain; the eight steps of a chain whose update
has an error gain >= 1/4 run on the policy's wide kernel sets: the first on "fp32", the next seven on "fp32h" - fp32 storage, fp16
operands, two weight passes).  Those eight steps are the FIRST eight of a chain: the warm-up (>= 16 steps: every kernel set's eager
step and graph capture) consumes them, so a timed window shorter than the rest of the chain (`--steps` < ~980, e.g. the driver's
--steps 20) contains NONE of them.  The line says how many it contained (`config.wide_set_steps_in_window`), times each wide set's
graph separately (`config.ms_per_step_wide_sets`) and, for such a window, reports the per-chain mix (992 fp16-set + 7 fp32h-set +
1 fp32-set steps) as `value`, the raw window rate as `config.value_window`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FULL_CFG = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                attn_res=[16], res_blocks=2, image_size=128)
SCHED = dict(schedule="cosine", n_timestep=1000, linear_start=1e-6, linear_end=1e-2)
GROUPS = 5                 # CAVE: 31 bands, n_subs=8, n_ovls=2 -> 5 spectral groups (AE.py:263)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, bf16 and fp16 alike, MI355X_MICROARCH.md
HEADLINE = "fp16"                # the fastest mode that meets north_star's tolerance (see `parity`)
DTYPE = {"bf16": "bf16 storage and MFMA operands (outside north_star's tolerance: experimental)",
         "fp16": "fp16 storage and MFMA operands, one weight pass; weights dithered over the chain's steps (fp16(w + d[k%4] ulp): the weight "
                 "rounding averages out instead of biasing the chain); the 8 steps of a chain with update gain >= 1/4 on the fp32 kernel set",
         "fp16x1": "fp16 (one weight pass)", "fp16x2": "fp16 (hi+lo fp16 weights wherever a kernel takes them)",
         "fp32": "fp32 storage, every product as three bf16 MFMAs (hi*hi + hi*lo + lo*hi)"}
HBM_PEAK_GBPS = 8000.0           # HBM3E spec (6290 measured by a streaming read), MI355X_MICROARCH.md


def build_model(dev, precision):
    from hsi_dmgasr_amd.init import init_weights_orthogonal
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                  attn_res=[16], res_blocks=2, dropout=0.2, image_size=128, precision=precision)
    init_weights_orthogonal(u, seed=0)                      # model/networks.py:110-112
    gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, loss_type="l1", conditional=True)
    gd = gd.to(dev).eval()
    gd.set_loss(dev)
    gd.set_new_noise_schedule(SCHED, dev)
    gd.noise, gd.seed = "philox", 2
    return gd


def conv_roofline(run, batch, reps=3, peak=MFMA_BF16_PEAK_TFLOPS, passes=1, mode=None):
    """Roofline of the DOMINANT kernel of one step.  Every conv launch of one eager step is bracketed with HIP events
    on the launch stream (best of `reps`); launches are grouped by the kernel the C ABI dispatches to
    (hsidm_conv_kernel_id) and the group with the largest total time is reported:
    achieved = sum of algorithmic FLOPs of its launches / sum of their durations."""
    from hsi_dmgasr_amd import ops
    best = None
    for _ in range(reps):
        recs = []
        ops.set_conv_probe(recs)
        run._enqueue()
        ops.set_conv_probe(None)
        torch.cuda.synchronize()
        for r in recs:
            r["ms"] = r["e0"].elapsed_time(r["e1"])
        if best is None:
            best = recs
        else:
            for a, b in zip(best, recs):
                a["ms"] = min(a["ms"], b["ms"])
    groups = {}
    for r in best:
        g = groups.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
        g["ms"] += r["ms"]; g["flops"] += r["flops"]; g["bytes"] += r["bytes"]; g["n"] += 1
    name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
    all_ms = sum(g["ms"] for g in groups.values())
    all_fl = sum(g["flops"] for g in groups.values())
    top = max((r for r in best if r["kernel"] == name), key=lambda r: r["flops"] / r["ms"])
    traffic = None
    # PMC passes (FETCH_SIZE x2 + WRITE_SIZE) of the same command, one file per precision mode (profiles/README.md)
    tf = os.path.join(ROOT, "profiles", "hbm_traffic_%s.json" % mode) if mode else os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tf):
        tab = json.load(open(tf))
        # (the LDS-tiled kernel takes its transform at run time: its counters are keyed by geometry only)
        traffic = tab.get(name) or tab.get(name.replace(" gn+silu", "").replace(" gn", ""))
        if traffic is not None and traffic.get("batch_per_gpu") not in (None, batch):
            traffic = None                                      # counters were collected at another batch: not this run's traffic
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12       # algorithmic FLOPs (the fp32 mode issues `passes` MFMAs per product)
    # the launch closest to the HBM roof (north-star: "fraction of the HBM roofline on the fused ResnetBlock kernel"): a 3x3
    # GN+SiLU conv of the 128x128 level; algorithmic bytes = activations in + out + weights, against the 8 TB/s spec
    hb = max((r for r in best if r["ksize"] == 3 and r["stride"] == 1 and "gn+silu" in r["kernel"]), key=lambda r: r["bytes"] / r["ms"])
    hbm_view = dict(kernel=hb["kernel"], cin=hb["cin"], cout=hb["cout"], hw=list(hb["hw"]), us=hb["ms"] * 1e3,
                    algorithmic_bytes=hb["bytes"], achieved_GBps=hb["bytes"] / (hb["ms"] * 1e-3) / 1e9, peak_GBps=HBM_PEAK_GBPS,
                    frac=hb["bytes"] / (hb["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    tflops=hb["flops"] / (hb["ms"] * 1e-3) / 1e12)
    # the same ResnetBlock under SURVEY 8(d)'s FUSED-unit byte definition (input read once, output written once, both convs'
    # weights; the intermediate h does not count) over the time of its two conv launches - GroupNorm 2 needs the global
    # statistics of h, so the block is two launches here and this fraction is about half of the two-kernel one
    pair = next(((a, b) for a, b in zip(best, best[1:]) if a["kernel"] == hb["kernel"] == b["kernel"] and
                 (a["cin"], a["cout"], b["cin"], b["cout"]) == (hb["cin"], hb["cout"], hb["cout"], hb["cout"]) and
                 tuple(a["hw"]) == tuple(hb["hw"]) == tuple(b["hw"])), None)
    if pair is not None:
        a, b = pair
        esz = 4 if passes > 1 else 2                             # bytes per stored element (fp32 mode: fp32 storage, hi + lo weights)
        px = batch * hb["hw"][0] * hb["hw"][1]
        fb = esz * px * (a["cin"] + b["cout"]) + esz * 9 * (a["cin"] * a["cout"] + b["cin"] * b["cout"])
        ft = (a["ms"] + b["ms"]) * 1e-3
        ff = a["flops"] + b["flops"]
        t_peak = ff / (peak * 1e12)                              # both convolutions at the nominal dense matrix peak, nothing else
        hbm_view["fused_unit"] = dict(definition="SURVEY 8(d) fused ResnetBlock: in + out + both weights, h not counted",
                                      algorithmic_bytes=fb, us=ft * 1e6, achieved_GBps=fb / ft / 1e9, frac=fb / ft / 1e9 / HBM_PEAK_GBPS,
                                      algorithmic_flops=ff, flop_per_byte=ff / fb, ridge_flop_per_byte=peak * 1e12 / (HBM_PEAK_GBPS * 1e9),
                                      frac_if_at_nominal_mfma_peak=fb / t_peak / 1e9 / HBM_PEAK_GBPS,
                                      note="matrix-bound block: its HBM fraction = its MFMA fraction x ridge / intensity (DESIGN.md section 6)")
    return dict(bound="mfma", hbm_view=hbm_view, achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, mfma_passes_per_product=passes,
                traffic=traffic, kernel=name, launches=dom["n"], avg_launch_us=dom["ms"] / dom["n"] * 1e3,
                algorithmic_flops_per_launch=dom["flops"] / dom["n"], algorithmic_bytes_per_launch=dom["bytes"] / dom["n"],
                share_of_conv_time=dom["ms"] / all_ms,
                best_launch=dict(cin=top["cin"], cout=top["cout"], hw=list(top["hw"]), us=top["ms"] * 1e3,
                                 tflops=top["flops"] / (top["ms"] * 1e-3) / 1e12),
                all_conv_kernels=dict(launches=len(best), flops_per_step=all_fl, ms_per_step=all_ms,
                                      tflops=all_fl / (all_ms * 1e-3) / 1e12),
                # every kernel instance of the conv family, largest first: the dominant one above is simply the first row
                # ("+proj": the launches that carry a ResnetBlock's 1x1 projection as one-tap chunks; its FLOPs are counted)
                kernels=[dict(kernel=k, launches=g["n"], ms_per_step=g["ms"], avg_launch_us=g["ms"] / g["n"] * 1e3,
                              tflops=g["flops"] / (g["ms"] * 1e-3) / 1e12, frac=g["flops"] / (g["ms"] * 1e-3) / 1e12 / peak)
                         for k, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])],
                # the conv_v3 launches of the 128x128 level (64 couts, with and without the fused projection) as ONE row: the kernel
                # FAMILY with the largest share of the step, whichever instance leads
                conv_v3_family=family_row(groups, "conv_v3 bn64", peak))


def family_row(groups, prefix, peak):
    rows = [g for k, g in groups.items() if k.startswith(prefix)]
    if not rows:
        return None
    ms, fl, n = sum(g["ms"] for g in rows), sum(g["flops"] for g in rows), sum(g["n"] for g in rows)
    return dict(kernel=prefix + " (all instances)", launches=n, ms_per_step=ms, avg_launch_us=ms / n * 1e3,
                tflops=fl / (ms * 1e-3) / 1e12, frac=fl / (ms * 1e-3) / 1e12 / peak)


def chain_parity(dev, modes=("fp16",), long_modes=(), orth_net=None):
    """north_star's tolerance, measured here on the chain fixture SET: the reference's own validation iteration (sr_gae.py:436-474)
    at its shipped configuration - T = 20 cosine chain, one CAVE image = 5 group latents, 97.8 M UNet, pretrained CAVE
    autoencoder - run BY THE IMPORTED REFERENCE for two weight sets (synthetic Gaussian keyed by name; the reference's own
    orthogonal initialisation = the weights this benchmark times) x two noise / cube draws, plus the metric's own chain length:
    the 1000-step loop on the orthogonal weights (tests/golden/chain.npz, tests/golden/chains/*.npz, make_golden_chain*.py).
    Per mode: every fixture's deviations, and the WORST over the fixtures - `meets_north_star` is about the worst.  Quality
    indices by the product's device kernels (hsidm_hsi_metrics, pinned to the reference's eval_hsi.py in the tests).  The
    autoencoder runs in its default fp32 mode."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from synth import CHAIN_CHIKUSEI, CHAIN_LONG_SET, CHAIN_SET, chain_cubes_draw, chain_noise_draw, chain_weights_check, synth_param
    from hsi_dmgasr_amd import gae, metrics, pipeline
    from hsi_dmgasr_amd.init import init_weights_orthogonal
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    gold = os.path.join(ROOT, "tests", "golden")
    m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=8, n_ovls=2, n_colors=31, n_feats=64, precision="fp32").to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(os.path.join(gold, "gae_cav_state.npz")).items()})
    m_chi = None
    if os.path.exists(os.path.join(gold, "gae_chi_state.npz")):      # BASELINE configs[2]: the pretrained Chikusei autoencoder (128 bands, G = 11)
        m_chi = gae.GAE(gae.Encoder, gae.Decoder, n_subs=16, n_ovls=4, n_colors=128, n_feats=64, precision="fp32").to(dev).eval()
        m_chi.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(os.path.join(gold, "gae_chi_state.npz")).items()})
    nets = {}

    def net(weights, g):
        if weights not in nets:
            u = unet.UNet(dropout=0.2, **FULL_CFG)
            if weights == "synth":
                u.load_state_dict({k: torch.from_numpy(synth_param("unet_full." + k, tuple(v.shape))) for k, v in u.state_dict().items()
                                   if not k.startswith("_")}, strict=False)
            else:       # build_model()'s weights; the fixture's probes prove they are the ones the reference's init_weights produced
                if orth_net is not None:
                    u = orth_net                 # the very network the headline was timed on
                else:
                    init_weights_orthogonal(u, seed=0)
                keys = [str(k) for k in g["w_keys"]]
                usd = u.state_dict()
                chain_weights_check({k: usd[k].detach().cpu() for k in keys}, keys, g["w_probe"])
            nets[weights] = u.to(dev).eval()
        return nets[weights]

    out = {"fixtures": "the reference's validation iteration (one CAVE image, 5 group latents 3x128x128, 97.8M UNet, pretrained CAVE "
                       "autoencoder, cosine schedule) as run by the imported reference: weights {synthetic, reference orthogonal init} x "
                       "three draws at T=20 (the last two generated after round 5's precision policy was fixed), two 1000-step chains (one per weight set), and a Chikusei image (128 bands, 11 group latents, pretrained Chikusei autoencoder: "
                       "BASELINE configs[2]; every 4th band of its cube compared); bounds 1e-3 relative / 0.01 dB / 0.001 deg (BASELINE.json north_star)"}
    per = {p: {} for p in modes}
    with torch.no_grad():
        for fx in tuple(CHAIN_SET) + (tuple(CHAIN_LONG_SET) if long_modes else ()) + ((("chi",) + tuple(CHAIN_CHIKUSEI),) if m_chi is not None else ()):
            chi = fx[0] == "chi"
            weights, draw, steps = fx[-3:]
            name = "chain.npz" if fx == ("synth", 0, 20) else os.path.join("chains", ("chi_" if chi else "") + "%s_n%d_T%d.npz" % (weights, draw, steps))
            if not os.path.exists(os.path.join(gold, name)):
                continue
            g = np.load(os.path.join(gold, name))
            gd = diffusion.GaussianDiffusion(net(weights, g), image_size=128, channels=3, conditional=True)
            gd.set_loss(dev)
            gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=steps, linear_start=1e-6, linear_end=1e-2), dev)
            hr, sr = chain_cubes_draw(draw, 128 if chi else 31)
            ngr = g["x0"].shape[0]
            x_T = G(np.concatenate([chain_noise_draw(draw, gi, 0) for gi in range(ngr)]))
            noise = G(np.stack([np.concatenate([chain_noise_draw(draw, gi, k) for gi in range(ngr)]) for k in range(1, steps)]))
            truth, ref_lat = G(hr), G(g["x0"])
            # (the Chikusei fixture stores every fourth band of the reference's cube and its indices of the whole cube)
            ref_y = G(g["y_sub4"]) if chi else G(g["y"])
            q_ref = (float(g["mpsnr_formula"]), float(g["sam_oracle"])) if chi else tuple(float(v) for v in metrics.quality_indices(truth, ref_y)[0][:2])
            # (the 1000-step chains only in `long_modes`; bf16 - outside the tolerance by 8x - not on the Chikusei fixture)
            for prec in ([p for p in modes if p in long_modes] if steps > 100 else ([p for p in modes if p != "bf16"] if chi else modes)):
                ae = m_chi if chi else m
                y, lat = pipeline.super_resolve(ae, gd, G(sr), x_T=x_T, noise=noise, precision=prec)
                q = metrics.quality_indices(truth, y)[0]
                e_lat = float((lat[0] - ref_lat).double().norm() / ref_lat.double().norm())
                # the same deviation over the elements the reference did NOT clamp to +-1 (diffusion.py:164): a saturated element carries no
                # error and all of its weight in the norm (38 % of the T = 20 reference latents are exactly +-1)
                free = ref_lat.abs() < 1.0
                e_unsat = float(((lat[0] - ref_lat)[free]).double().norm() / ref_lat[free].double().norm())
                ys = y[:, ::4] if chi else y
                e_y = float((ys - ref_y).double().norm() / ref_y.double().norm())
                # the CONTINUOUS companion of the SAM index: both latent sets through the same fp32-mode decoder WITHOUT the final clamp
                # (sr_gae.py:473-474) - no spectrum is exactly zero there, so the index has no support to flip
                C = 128 if chi else 31
                raw_ours, raw_ref = ae.decode_batched(lat, C), ae.decode_batched(ref_lat.view(1, ngr, 3, 128, 128), C)
                row = dict(latents_rel_err=e_lat, latents_rel_err_unsaturated=e_unsat, saturated_latent_fraction=float(1.0 - free.double().mean()),
                           cube_rel_err=e_y, dPSNR_dB=abs(float(q[0]) - q_ref[0]), dSAM_deg=abs(float(q[1]) - q_ref[1]),
                           dSAM_unclamped_deg=abs(sam_unclamped(truth, raw_ours) - sam_unclamped(truth, raw_ref)))
                if not chi:     # (the Chikusei fixture stores a quarter of the reference's cube: no per-pixel comparison of the SAM support)
                    row["sam_support_flips"], row["dSAM_common_support_deg"], row["flipped_pixel_norm_over_cube_rms"] = sam_support(truth, y, ref_y)
                # the value the SAM bound is applied to: the reference's strict index, unless the two cubes disagree about which pixels it
                # skips AND the disagreement is the index's discontinuity and nothing else (<= 2 pixels, each at the clamp boundary in both cubes)
                fl = row.get("sam_support_flips") or 0
                boundary = 0 < fl <= SAM_MAX_FLIPS and row["flipped_pixel_norm_over_cube_rms"] <= SAM_FLIP_NORM
                row["dSAM_gated_deg"], row["sam_gate_is_strict_index"] = (row["dSAM_common_support_deg"], False) if boundary else (row["dSAM_deg"], True)
                per[prec][":".join(str(v) for v in fx).replace(":%d:%d" % (draw, steps), ":n%d:T%d" % (draw, steps))] = row
            del noise, gd
    for prec in modes:
        rows = per[prec]
        if not rows:
            continue
        worst = {k: max(r[k] for r in rows.values()) for k in ("latents_rel_err", "latents_rel_err_unsaturated", "cube_rel_err", "dPSNR_dB", "dSAM_deg",
                                                                 "dSAM_unclamped_deg")}
        cont = bool(worst["latents_rel_err"] <= 1e-3 and worst["latents_rel_err_unsaturated"] <= 1e-3 and worst["cube_rel_err"] <= 1e-3 and
                    worst["dPSNR_dB"] <= 0.01 and worst["dSAM_unclamped_deg"] <= 1e-3)
        sam_gate = max(r["dSAM_gated_deg"] for r in rows.values())
        out[prec] = dict(worst, fixtures=rows, n_fixtures=len(rows), sam_support_flips=sum(r.get("sam_support_flips") or 0 for r in rows.values()),
                         strict_sam_misses=sorted(k for k, r in rows.items() if r["dSAM_deg"] > 1e-3),
                         dSAM_deg_on_common_support=sam_gate,
                         meets_north_star=bool(cont and worst["dSAM_deg"] <= 1e-3),
                         meets_north_star_with_sam_on_common_support=bool(cont and sam_gate <= 1e-3))
    del nets, m, m_chi
    torch.cuda.empty_cache()
    return out


SAM_MAX_FLIPS, SAM_FLIP_NORM = 2, 1e-3          # (tests/helpers.py: sam_gate - the same two conditions)


def sam_support(truth, y, ref):
    """(pixels whose zero-spectrum membership differs between our cube and the reference's, |SAM(truth, ours) - SAM(truth, reference's)|
    in degrees over the pixels BOTH keep, the largest spectrum norm of a flipped pixel in either cube over the reference cube's rms).
    The reference's SAM (eval_hsi.py:47-65) skips pixels whose predicted spectrum is exactly zero: the index is discontinuous where a
    spectrum sits at the clamp(0, 1) boundary of the decoded cube - one pixel entering or leaving the mean moves it by
    (its angle - mean) / N, about 1.4e-3 degrees on these 128 x 128 cubes, however small the deviation that flipped it.  `parity`
    reports the strict index, the number of such pixels, the index on the common support, and applies the bound to the latter ONLY
    when at most SAM_MAX_FLIPS pixels flipped and each has a spectrum norm <= SAM_FLIP_NORM x the cube's rms in both cubes."""
    import math
    t, a, b = (v[0].reshape(v.shape[1], -1).float() for v in (truth, y, ref))
    nt, na, nb = t.norm(dim=0), a.norm(dim=0), b.norm(dim=0)
    flip = (na != 0) != (nb != 0)
    flips = int(flip.sum())
    ok = (nt != 0) & (na != 0) & (nb != 0)
    sam = lambda p, n: float(torch.arccos(((t * p).sum(dim=0)[ok] / (nt[ok] * n[ok])).clamp(-1.0, 1.0)).double().mean()) * 180.0 / math.pi
    worst = float(torch.maximum(na[flip], nb[flip]).max() / b.double().pow(2).mean().sqrt()) if flips else 0.0
    return flips, abs(sam(a, na) - sam(b, nb)), worst


def sam_unclamped(truth, raw):
    """Mean spectral angle in degrees over every pixel with a non-zero true spectrum, for an UN-CLAMPED decoded cube [1, C, H, W]."""
    import math
    t, p = (v[0].reshape(v.shape[1], -1).double() for v in (truth, raw))
    nt, npn = t.norm(dim=0), p.norm(dim=0)
    ok = (nt != 0) & (npn != 0)
    return float(torch.arccos(((t * p).sum(dim=0)[ok] / (nt[ok] * npn[ok])).clamp(-1.0, 1.0)).mean()) * 180.0 / math.pi


def cpu_baseline(cases=((1, 42), (5, 12)), warm=2, segments=3, sd=None):
    """The oracle on the host CPU: full-size UNet p_sample steps (fp32), a bounded sample (~10 s in total) of the same workload
    at B = 1 (one group latent per call: how the reference itself runs, sr_gae.py:458-465 loops over the groups at batch 1) and
    B = 5 (one CAVE image's five group latents in one batch): `steps` of the 1000 reverse steps each (every step costs the
    same).  Threads are pinned to the usable cores; each case is timed in `segments` equal parts and reports the
    MEDIAN part (hosts of this pool are shared: single samples ranged 7.7 ... 14.7).  `value` is the B = 1 case - the
    reference's own mode of execution; both are in `cases`."""
    from oracle import diffusion as odiff, sr3_unet
    from hsi_dmgasr_amd.init import init_weights_orthogonal
    from hsi_dmgasr_amd.sr3_modules import unet
    threads = min(usable_cpus(), 64)
    torch.set_num_threads(threads)
    if sd is None:          # (main() passes the timed network's own state: the same orthogonal-init weights, one initialisation fewer)
        u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                      attn_res=[16], res_blocks=2, dropout=0.2, image_size=128)
        init_weights_orthogonal(u, seed=0)
        sd = {k: v.detach() for k, v in u.state_dict().items()}
    sched = odiff.noise_schedule(SCHED)
    den = lambda xx, gam: sr3_unet.unet_forward(sd, FULL_CFG, xx, gam)
    recs = {}
    for batch, steps in cases:
        g = torch.Generator().manual_seed(1)
        cond = torch.randn((batch, 3, 128, 128), generator=g).clamp(-2.5, 2.5)
        x = torch.randn((batch, 3, 128, 128), generator=g)
        per = max(1, steps // segments)
        parts = []
        with torch.no_grad():
            for k in range(warm):
                x = odiff.p_sample_step(den, sched, x, cond, 999 - k, torch.randn(x.shape, generator=g))
            k = warm
            for _ in range(segments):
                t0 = time.perf_counter()
                for _ in range(per):
                    x = odiff.p_sample_step(den, sched, x, cond, 999 - k, torch.randn(x.shape, generator=g))
                    k += 1
                parts.append(time.perf_counter() - t0)
        med = sorted(parts)[len(parts) // 2]
        recs["batch_%d" % batch] = dict(value=per * batch / med, s_per_step=med / per, steps=per * segments, seconds=sum(parts),
                                        segment_rates=[per * batch / p for p in parts])
    head = recs["batch_%d" % cases[0][0]]
    return dict(value=head["value"], unit="denoise-steps*batch/s", cores=threads, kind="port", cases=recs,
                sample="p_sample steps of the full 97.8M UNet on the fp32 oracle, %d pinned threads, median of %d segments: %s" % (
                    threads, segments, "; ".join("%d steps at batch %s in %.1f s" % (r["steps"], k.split("_")[1], r["seconds"])
                                                 for k, r in recs.items())))


def log(msg):
    if os.environ.get("HSIDM_BENCH_VERBOSE"):
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def usable_cpus():
    """CPUs this process may really use: affinity mask and cgroup quota, not the machine's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def gae_bytes(bands, n_subs, groups, esz, feats=64, tfeats=32, hw=128 * 128):
    """HBM bytes of one patch through the group-autoencoder, two ways (these 64-channel maps are HBM-bound, 144-288 FLOP/B):
    `unit` - the fused-unit minimum: every unit (head conv, ResBlock, ResAttentionBlock, final conv, overlap-average) reads its
    input once and writes its output once, intermediates inside a unit do not count; `launch` - what this build's launches move
    (a ResBlock is two 3x3 launches with h and the residual in between, a ResAttentionBlock two 1x1 launches + the scale /
    residual pass that has to wait for the global average).  esz: bytes per stored activation (4 in the fp32 mode)."""
    pad8 = lambda c: (c + 7) // 8 * 8
    F, S = feats, esz

    def branch(cin, f, nblk):            # head + nblk x (ResBlock + ResAttentionBlock), in stored elements per pixel: (unit, launch)
        # launch level, per block: ResBlock 5 f (conv: in, out; conv: in, residual, out) + ResAttentionBlock 7 f (1x1: in, out; 1x1: in, out;
        # scale + residual pass: 3) - or 5 f where the 1x1 pair is ONE launch (64 features: hsidm_conv1x1_pair, round 6)
        per = 10 if f == 64 else 12
        return (pad8(cin) + f) + nblk * 4 * f, (pad8(cin) + f) + nblk * per * f + f
    eu, el = branch(n_subs, F, 3)
    enc_unit = groups * hw * (n_subs * 4 + pad8(n_subs) * S + eu * S + F * S + 3 * 4)
    enc_launch = groups * hw * (n_subs * 4 + pad8(n_subs) * S + el * S + F * S + 3 * 4)
    du, dl = branch(3, F, 3)
    tu, tl = branch(bands, tfeats, 2)
    common = groups * hw * n_subs * 4 + hw * bands * 4 + hw * (bands * 4 + pad8(bands) * S) + hw * (tfeats * S + 2 * bands * 4)
    dec_unit = groups * hw * (3 * 4 + 8 * S + du * S + F * S + n_subs * 4) + common + hw * tu * S
    dec_launch = groups * hw * (3 * 4 + 8 * S + dl * S + F * S + n_subs * 4) + common + hw * tl * S
    return dict(encode=dict(unit=enc_unit, launch=enc_launch), decode=dict(unit=dec_unit, launch=dec_launch))


def gae_bench(dev, patches, reps=5, bands=31, n_subs=8, n_ovls=2, groups=5, flops=None):
    """Group-autoencoder encode / decode of `patches` cubes of `bands` x 128 x 128 (G spectral groups stacked on the batch axis),
    pretrained-checkpoint architecture (CAVE: n_subs 8, n_ovls 2; Chikusei: 16 / 4, G = 11; 64 features) in every
    precision mode, with each 16-bit mode's deviation from the fp32 mode on the same cubes (PSNR of the reconstruction against the
    input: the autoencoder's decode sets the final PSNR, so its mode is held to the 0.01 dB bound too; weights: the module's
    default initialisation, there is no pretrained Chikusei-size checkpoint in reach of the GPU box)."""
    from hsi_dmgasr_amd import gae, metrics
    out = {}
    flops = flops or dict(encode=41.3e9, decode=41.3e9 + 1.9e9)  # per patch at 128 x 128 (decode includes the trunk)
    x = torch.rand(patches, bands, 128, 128, generator=torch.Generator().manual_seed(3))
    x = ((x[:, :-2] + x[:, 1:-1] + x[:, 2:]) / 3.0 if bands > 2 else x)                    # smooth along the band axis
    x = torch.nn.functional.pad(x, (0, 0, 0, 0, 1, 1), mode="replicate")[:, :bands].contiguous().to(dev)
    sd, ref_q = None, None
    for prec in ("fp32", "fp16"):
        m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=n_subs, n_ovls=n_ovls, n_colors=bands, n_feats=64, precision=prec).to(dev).eval()
        if sd is None:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
        else:
            m.load_state_dict(sd)
        z = m.encode_batched(x)
        q = metrics.quality_indices(x, m.decode_batched(z, bands).clamp(0.0, 1.0))[:, :2].double().mean(dim=0)
        if ref_q is None:
            ref_q = q
        rec = {}
        for name, fn in (("encode", lambda: m.encode_batched(x)), ("decode", lambda: m.decode_batched(z, bands))):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e30
            for _ in range(reps):
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            tf = flops[name] * patches / (best * 1e-3) / 1e12
            rec[name + "_ms"] = best
            rec[name + "_tflops"] = tf
            rec[name + "_frac_of_mfma_peak"] = tf / MFMA_BF16_PEAK_TFLOPS
            by = gae_bytes(bands, n_subs, groups, 4 if prec == "fp32" else 2)[name]
            # HBM axis: algorithmic bytes under the fused-unit definition, and the bytes this build's launches move, per second
            rec[name + "_hbm"] = dict(unit_bytes=by["unit"] * patches, launch_bytes=by["launch"] * patches,
                                      unit_GBps=by["unit"] * patches / (best * 1e-3) / 1e9, launch_GBps=by["launch"] * patches / (best * 1e-3) / 1e9,
                                      unit_frac_of_hbm=by["unit"] * patches / (best * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                      launch_frac_of_hbm=by["launch"] * patches / (best * 1e-3) / 1e9 / HBM_PEAK_GBPS)
        rec["dPSNR_dB_vs_fp32_mode"] = abs(float(q[0] - ref_q[0]))
        rec["dSAM_deg_vs_fp32_mode"] = abs(float(q[1] - ref_q[1]))
        rec["within_0.01dB_0.001deg"] = bool(rec["dPSNR_dB_vs_fp32_mode"] <= 0.01 and rec["dSAM_deg_vs_fp32_mode"] <= 1e-3)
        out[prec] = rec
        del m, z
    out["patches"] = patches
    out["cube"] = "%dx128x128, G=%d" % (bands, groups)
    return out


def train_bench(dev, batch=4, steps=10, warm=3, precisions=("bf16", "fp32"), reduce_grads=False):
    """BASELINE configs[4]: one joint-train step of the UNet (p_losses forward in training mode with Dropout 0.2, hand-written
    backward, fused Adam, re-pack) on `batch` GAE latents of 3 x 128 x 128 per GPU; model/model.py:49-59.  Returns per precision
    mode the time per step (HIP events around whole steps) and its split into forward+loss / backward / optimiser."""
    from hsi_dmgasr_amd.init import init_weights_orthogonal
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    out = {"batch_per_gpu": batch, "steps": steps}
    g = torch.Generator().manual_seed(5)
    data = {"HR": torch.randn((batch, 3, 128, 128), generator=g).clamp(-2.5, 2.5).to(dev),
            "SR": torch.randn((batch, 3, 128, 128), generator=g).clamp(-2.5, 2.5).to(dev)}
    for prec in precisions:
        u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                      attn_res=[16], res_blocks=2, dropout=0.2, image_size=128, precision=prec)
        init_weights_orthogonal(u, seed=0)
        gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, loss_type="l1", conditional=True).to(dev).train()
        gd.set_loss(dev)
        gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=20, linear_start=1e-6, linear_end=1e-2), dev)   # config/sr_sr3_16_128.json:96-101
        tr = gd.trainer(lr=1e-5)                                                                                        # :120-123
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t_f = t_b = t_o = 0.0
        losses = []
        for i in range(warm + steps):
            if i == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            ev[0].record()
            loss, state = tr.forward_loss(data)
            ev[1].record()
            tr.backward_loss(state, 1.0 / float(state[2]))
            ev[2].record()
            tr.optimizer_step()
            ev[3].record()
            if i >= warm:
                torch.cuda.synchronize()
                t_f += ev[0].elapsed_time(ev[1]); t_b += ev[1].elapsed_time(ev[2]); t_o += ev[2].elapsed_time(ev[3])
                losses.append(float(loss) / float(state[2]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert all(l == l and abs(l) < 1e6 for l in losses), "training diverged"
        out[prec] = dict(ms_per_step=dt / steps * 1e3, value=steps * batch / dt, unit="train-steps*batch/s",
                         forward_ms=t_f / steps, backward_ms=t_b / steps, optimizer_ms=t_o / steps,
                         l_pix_first=losses[0], l_pix_last=losses[-1],
                         tflops=3 * 92.35e9 * batch / (dt / steps) / 1e12)      # fwd + dgrad + wgrad ~ 3x the forward's 92.35 GFLOP per sample
        # the same step as ONE hipGraph replay (Trainer.optimize_parameters from its third call on)
        for _ in range(3):
            tr.optimize_parameters(data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.optimize_parameters(data)
        torch.cuda.synchronize()
        dg = time.perf_counter() - t0
        assert bool(torch.isfinite(loss))
        out[prec]["graph_ms_per_step"] = dg / steps * 1e3
        out[prec]["graph_value"] = steps * batch / dg
        out[prec]["graph_tflops"] = 3 * 92.35e9 * batch / (dg / steps) / 1e12
        del tr, gd, u
        torch.cuda.empty_cache()
    return out


def flush_c_stdio():
    """RCCL prints a version banner through C stdio when its first communicator comes up; left in the C buffer it would be
    flushed at process exit, i.e. AFTER rank 0's JSON line.  Flushing every rank's C buffers right after the first collective
    keeps the JSON line the last line of the job's stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell: re-run under torch.distributed.run as a CHILD process (never exec:
    nothing here has touched the GPU yet, and nothing will in this parent)."""
    port = os.environ.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main_train(args, dev, rank, world, use_dist):
    """--workload train: BASELINE configs[4] as its own benchmark line (one rank per GPU, gradients averaged over RCCL inside the
    backward pass).  K timed steps after W warm-up steps, barrier + synchronize on both sides, MAX over ranks."""
    import torch.distributed as dist
    from hsi_dmgasr_amd import parallel
    from hsi_dmgasr_amd.init import init_weights_orthogonal
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    # the training step runs in the bf16 or the fp32 mode (the fp16 mode is an inference mode: DESIGN.md section 5)
    prec, B = (args.precision if args.precision in ("bf16", "fp32") else "bf16"), args.train_batch
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                  attn_res=[16], res_blocks=2, dropout=0.2, image_size=128, precision=prec)
    init_weights_orthogonal(u, seed=0)
    gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, loss_type="l1", conditional=True).to(dev).train()
    gd.set_loss(dev)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=20, linear_start=1e-6, linear_end=1e-2), dev)
    parallel.broadcast_module_(gd, src=0)
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
        flush_c_stdio()
    tr = gd.trainer(lr=1e-5)
    g = torch.Generator().manual_seed(5 + rank)
    data = {"HR": torch.randn((B, 3, 128, 128), generator=g).clamp(-2.5, 2.5).to(dev),
            "SR": torch.randn((B, 3, 128, 128), generator=g).clamp(-2.5, 2.5).to(dev)}
    steps = min(args.steps, 200)
    for _ in range(max(args.warmup, 2)):
        tr.optimize_parameters(data)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.optimize_parameters(data)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    el = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    dt = float(el.item())
    assert torch.isfinite(loss), "training diverged"
    if rank == 0:
        print(json.dumps({
            "metric": "UNet joint-train steps/sec x batch (forward + backward + Adam), GAE latents 3x128x128, BASELINE configs[4]",
            "value": steps * B * world / dt, "unit": "train-steps*batch/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": prec if prec == "bf16" else "fp32 (bf16x3 split forward/dgrad, exact fp32 MFMA wgrad)",
            "data": "synthetic (orthogonal-init weights seed 0, N(0,1) latents clipped to +-2.5)",
            "config": {"workload": "sr_gae.py joint-train step: SR3 UNet 97.8M fwd+bwd (Dropout 0.2, L1) + Adam lr 1e-5, GAE frozen",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "grad_allreduce": "bucketed RCCL all-reduce of the flat fp32 gradient buffer (%d MB), launched from inside the backward pass"
                                         % (tr.grad.numel() * 4 >> 20)},
            "rccl_ranks": dist.get_world_size() if use_dist else 1, "l_pix": float(loss),
            "tflops": 3 * 92.35e9 * B * world / (dt / steps) / 1e12}))
    if use_dist:
        dist.destroy_process_group()


def _r(v, nd=4):
    """Numbers of the compact line: `nd` significant digits are what the measurement carries."""
    if isinstance(v, float):
        return float("%.*g" % (nd + 2, v))
    return v


def compact_line(head, roof=None, parity=None, cpu=None, mode=HEADLINE, detail_path=None, extra=None):
    """The contract line (the LAST stdout line of rank 0) from the measured objects - kept under 4 KB so that a driver reading
    the tail of stdout always gets the whole object (tests/test_host_logic.py holds it to that).  `head`: the contract's scalar
    fields + config; roof / parity / cpu: the full objects (conv_roofline, chain_parity, cpu_baseline), of which only the
    headline figures are kept here - the full objects go to the detail file."""
    line = dict(head)
    if roof is not None:
        tr = roof.get("traffic")
        hv = roof.get("hbm_view") or {}
        fam = roof.get("conv_v3_family")
        line["roofline"] = {
            "bound": roof["bound"], "kernel": roof["kernel"], "launches": roof["launches"], "avg_launch_us": _r(roof["avg_launch_us"]),
            "algorithmic_flops_per_launch": _r(roof["algorithmic_flops_per_launch"]),
            "algorithmic_bytes_per_launch": _r(roof["algorithmic_bytes_per_launch"]),
            "achieved": _r(roof["achieved"]), "peak": roof["peak"], "unit": roof["unit"], "frac": _r(roof["frac"]),
            # HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE counter passes of this command (profiles/, same batch), or null
            "traffic": None if tr is None else _r(tr["hbm_bytes_per_launch"]),
            "traffic_source": None if tr is None else roof.get("traffic_source"),
            "share_of_conv_time": _r(roof["share_of_conv_time"]),
            "whole_step": None if roof.get("whole_step") is None else {k: _r(v) for k, v in roof["whole_step"].items()},
            "conv_v3_family": None if fam is None else {k: _r(v) for k, v in fam.items()},
            "fused_resnetblock_hbm_frac": _r((hv.get("fused_unit") or {}).get("frac")),
            "resnetblock_launch_hbm_frac": _r(hv.get("frac")),
        }
    if cpu is not None:
        line["cpu_baseline"] = {"value": _r(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                "sample": cpu["sample"],
                                "cases": {k: {"value": _r(r["value"]), "s_per_step": _r(r["s_per_step"])} for k, r in cpu["cases"].items()}}
    if parity is not None and parity.get(mode):
        pm = parity[mode]
        line["parity"] = {"mode": mode, "latents_rel_err": _r(pm["latents_rel_err"]),
                          # over the elements the reference did not clamp to +-1 (diffusion.py:164), and the SAM angle on the un-clamped cubes
                          "latents_rel_err_unsaturated": _r(pm.get("latents_rel_err_unsaturated")), "cube_rel_err": _r(pm["cube_rel_err"]),
                          "dPSNR_dB": _r(pm["dPSNR_dB"]), "dSAM_deg": _r(pm["dSAM_deg"]), "dSAM_unclamped_deg": _r(pm.get("dSAM_unclamped_deg")),
                          "strict_sam_misses": pm.get("strict_sam_misses"), "n_fixtures": pm["n_fixtures"],
                          "worst_of": sorted(pm["fixtures"]), "bounds": "1e-3 rel / 0.01 dB / 0.001 deg (north_star)",
                          "oracle": "chains run by the imported reference (tests/golden/chain*.npz)",
                          "meets_north_star": pm["meets_north_star"],
                          # the SAM index skips exactly-zero spectra (eval_hsi.py:57): pixels at that discontinuity, and the bound with the
                          # index taken where both cubes keep the same pixels (bench.py: sam_support)
                          "sam_support_flips": pm.get("sam_support_flips"), "dSAM_deg_on_common_support": _r(pm.get("dSAM_deg_on_common_support")),
                          "meets_north_star_with_sam_on_common_support": pm.get("meets_north_star_with_sam_on_common_support")}
        line["meets_north_star"] = pm["meets_north_star"]
    if extra:
        line.update(extra)
    if detail_path:
        line["detail"] = detail_path
    return line


def chain_mix(window_ms, steps, others, T, total_batch):
    """The per-chain mix of the measured step times.  window_ms: the timed window of `steps` steps; others: one dict per kernel set of the
    chain other than its base set - {"mode", "per_chain" (steps of a chain on it), "in_window" (steps of the window on it), "ms" (one step
    on it, timed separately)} - for the "fp16" policy the fp32 set (the first step of a chain) and the fp32h set (the next seven).
    Returns the dict bench.py puts into `chain_mix` / `config`, or None when a set could not be timed or the window held no base-set step."""
    n_in = sum(o["in_window"] for o in others)
    if any(o["ms"] is None for o in others) or steps <= n_in:
        return None
    t_base = (window_ms - sum(o["in_window"] * o["ms"] for o in others)) / (steps - n_in)
    n_pc = sum(o["per_chain"] for o in others)
    t_pc = sum(o["per_chain"] * o["ms"] for o in others)
    ms_mix = ((T - n_pc) * t_base + t_pc) / T
    # the same policy on the chain length the reference SHIPS (config/sr_sr3_16_128.json:98,104: n_timestep 20 for validation): the
    # high-gain steps are the first ones of a chain whatever its length, so a 20-step chain pays all of them in 20 steps (steady-state
    # step times; a chain's graph captures are not in it)
    t20 = t_pc + (20 - n_pc) * t_base if n_pc <= 20 else None
    return dict(ms_per_step_base_mode=t_base, chain_steps=T, steps_per_chain_other_modes=n_pc,
                other_modes=[dict(mode=o["mode"], steps_per_chain=o["per_chain"], steps_in_window=o["in_window"], ms_per_step=o["ms"]) for o in others],
                ms_per_step_chain_mix=ms_mix, value_chain_mix=total_batch / (ms_mix * 1e-3),
                value_T20=None if t20 is None else total_batch * 20 / (t20 * 1e-3))


def value_is_mix(mix, others, steps, T):
    """Is `value` the per-chain mix?  Yes when the window held fewer steps of some other set than that set's share of a chain (the
    driver's 20 steps hold none of the eight); a window that is a whole chain IS the mix and is reported as measured."""
    return mix is not None and any(o["in_window"] * T < o["per_chain"] * steps for o in others)


def timed_mode_replays(run, mode, n):
    """Average ms of `n` replays of the run's captured graph of `mode` (after the timed region; the state just keeps walking)."""
    g = run.graphs.get(mode)
    if g is None:
        # a set the window and the warm-up ran at most once (the fp32 set is ONE step per chain): captured here, the way ReverseRun.step does
        if not (run.fused and run.gd.use_graph):
            return None
        if mode not in run._eager_done:
            run._enqueue(mode)
            run._eager_done.add(mode)
        torch.cuda.synchronize()
        g = run.graphs[mode] = torch.cuda.CUDAGraph()
        if run._pool is None:
            run._pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(g, pool=run._pool):
            run._enqueue(mode)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--patches", type=int, default=48,
                    help="CAVE patches per GPU (x5 spectral groups = batch of 240 latents); throughput saturates around here: "
                         "120 latents -3 %, 640 latents +2 % (sweep in DESIGN.md)")
    ap.add_argument("--total-patches", type=int, default=0,
                    help="strong scaling: this many patches in total, sharded contiguously over the ranks (BASELINE configs[3]: 64)")
    ap.add_argument("--precision", default=HEADLINE, choices=["bf16", "fp32", "fp16", "fp16x1", "fp16x2"],
                    help="precision mode of the headline number (default: the fastest mode that meets north_star's tolerance)")
    ap.add_argument("--detail", action="store_true",
                    help="also measure the secondary legs into the detail file: other precision modes, small batches, the group "
                         "autoencoder, the training step, parity in every mode incl. the 1000-step reference chains (adds ~1 min)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"))
    ap.add_argument("--no-parity-long", action="store_true", help="parity without the two 1000-step reference chains (~8 s; -m gpu runs them too)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    # (legs of --detail, individually switchable)
    ap.add_argument("--no-modes", action="store_true", help="--detail without the bf16_mode / fp32_mode objects")
    ap.add_argument("--no-fp32", action="store_true", help="--detail without the fp32-mode object")
    ap.add_argument("--no-gae", action="store_true", help="--detail without the group-autoencoder object")
    ap.add_argument("--no-small", action="store_true", help="--detail without the small-batch object (40 and 5 latents per GPU)")
    ap.add_argument("--no-train", action="store_true", help="--detail without the training-step object (BASELINE configs[4])")
    ap.add_argument("--workload", default="sample", choices=["sample", "train"],
                    help="sample: the headline metric (reverse-diffusion steps); train: BASELINE configs[4], one joint-train step "
                         "(UNet forward + backward + Adam, gradients all-reduced over the ranks) on --train-batch latents per GPU")
    ap.add_argument("--train-batch", type=int, default=4, help="latents per GPU and training step (sr_gae.py:182 uses 4)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    torch.set_num_threads(min(usable_cpus(), 64))      # the host may expose 256 CPUs behind a 16-CPU quota
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("HSIDM_FORCE_DIST")        # FORCE: exercise the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    from hsi_dmgasr_amd import parallel, precision as P
    if args.detail or args.precision in P.EXPERIMENTAL_MODES or args.workload == "train":
        P.allow_experimental(True)          # the bf16 kernel set is measured beside the public modes, labelled `meets_north_star: false`
    if args.workload == "train":
        return main_train(args, dev, rank, world, use_dist)
    log('building model')
    gd = build_model(dev, args.precision)
    log('model on device')
    parallel.broadcast_module_(gd, src=0)                  # one-time weight broadcast over xGMI
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
        flush_c_stdio()
    if args.total_patches:
        lo, hi = parallel.shard_range(args.total_patches, rank, world)
        patches, total_patches, scaling = hi - lo, args.total_patches, "strong"
    else:
        patches, total_patches, scaling = args.patches, args.patches * world, "weak"
    if patches <= 0:
        raise SystemExit("rank %d has no patches (--total-patches %d over %d ranks)" % (rank, args.total_patches, world))
    batch = patches * GROUPS
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    cond = torch.randn((batch, 3, 128, 128), generator=g).clamp(-2.5, 2.5).to(dev)   # GAE latent range
    run = gd.make_run(cond, wrap=True)

    allgather_ms = None
    with torch.no_grad():
        # the chain's first steps run on the fp32 kernel set (eager step, graph capture, replays), then every dither phase of the fp16
        # set takes an eager step and a capture: behind WARM_MIN steps every step, timed or not, is a graph replay
        from hsi_dmgasr_amd.precision import family
        n_hi = sum(1 for m in run.modes if family(m) != family(run.modes[-1]))
        n_warm = max(args.warmup, n_hi + 2 * len({m for m in run.modes if family(m) == family(run.modes[-1])}))
        for _ in range(n_warm):
            run.step()
        torch.cuda.synchronize()
        log('warmup done')
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        el = torch.tensor([dt], dtype=torch.float64, device=dev)
        el_min = el.clone()
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            dist.all_reduce(el_min, op=dist.ReduceOp.MIN)        # a straggler shows as max >> min
        dt, dt_min = float(el.item()), float(el_min.item())
        assert torch.isfinite(run.x).all(), "sampler state diverged"
        log('timed region done: %.3f s' % dt)
        # steps of the timed window that ran in another mode than the chain's base mode (the precision schedule's fp32-mode steps
        # sit at the START of each chain: a window shorter than the rest of the chain has none), and the per-chain mix
        T = run_T = run.T
        base_mode = family(run.modes[-1])
        n_other = sum(1 for k in range(n_warm, n_warm + args.steps) if family(run.modes[k % T]) != base_mode)
        others = []
        for om in dict.fromkeys(m for m in run.modes if family(m) != base_mode):      # (in chain order: "fp32", then "fp32h")
            others.append(dict(mode=om, per_chain=sum(1 for m in run.modes if m == om),
                               in_window=sum(1 for k in range(n_warm, n_warm + args.steps) if run.modes[k % T] == om), ms=None))
        mix = None
        if others and rank == 0:
            for o in others:
                o["ms"] = timed_mode_replays(run, o["mode"], 8)
            mix = chain_mix(dt * 1e3, args.steps, others, T, total_patches * GROUPS)
        if use_dist:
            # the path's one data collective: every rank ends with all SR cubes (here: cube-sized stand-ins for the decoded
            # patches, 31 x 128 x 128 fp32 each = 2.0 MB per patch)
            cubes = torch.zeros((patches, 31, 128, 128), dtype=torch.float32, device=dev)
            parallel.all_gather_patches(cubes, total_patches)            # warm-up (communicator set-up)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            full = parallel.all_gather_patches(cubes, total_patches)
            torch.cuda.synchronize()
            allgather_ms = (time.perf_counter() - t1) * 1e3
            assert full.shape[0] == total_patches
            del cubes, full

        roof = None
        if rank == 0 and not args.no_roofline:
            roof = conv_roofline(run, batch, mode=args.precision)
            # (fp16 mode: launches with hi + lo weights issue 1.5 matrix passes per product; `achieved` counts the algorithmic FLOPs once)
            roof["mfma_passes_per_product"] = 2 if args.precision == "fp16x2" else 1
            roof["traffic_source"] = "profiles/hbm_traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; 2*FETCH+WRITE, gfx950)" % args.precision
            # the whole step against the same roof: SURVEY 8(d)'s 92.35 GFLOP per latent and step over the measured step time
            ms_base = mix["ms_per_step_base_mode"] if mix else dt / args.steps * 1e3
            tf = 92.35e9 * batch / (ms_base * 1e-3) / 1e12
            roof["whole_step"] = dict(algorithmic_flops=92.35e9 * batch, ms=ms_base, tflops=tf, frac=tf / roof["peak"])
    log('roofline done')

    def other_mode(prec, passes):
        """The same step, workload and batch in another precision mode (rank 0, N = 1)."""
        with torch.no_grad():
            r = gd.make_run(cond, wrap=True, precision=prec)
            n = max(10, min(50, args.steps // 20))
            for _ in range(n_warm):                             # (every kernel set of a scheduled chain captured before the clock starts)
                r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                r.step()
            torch.cuda.synchronize()
            d = time.perf_counter() - t0
            assert torch.isfinite(r.x).all()
            rec = dict(value=n * batch / d, unit="denoise-steps*batch/s", ms_per_step=d / n * 1e3, steps=n, dtype=DTYPE[prec],
                       roofline=None if args.no_roofline else conv_roofline(r, batch, reps=2, passes=passes, mode=prec))
            del r
        torch.cuda.empty_cache()
        return rec

    solo = rank == 0 and world == 1
    del run
    torch.cuda.empty_cache()
    fp32 = bf16 = None
    if solo and args.detail and not args.no_modes and args.precision == HEADLINE:
        bf16 = other_mode("bf16", 1)
        log('bf16 mode done')
        if not args.no_fp32:
            fp32 = other_mode("fp32", 3)      # fp32 storage, every product as three bf16 MFMAs
            log('fp32 mode done')
    parity = None
    if solo and not args.no_parity:
        # default: the headline mode over every reference chain (six CAVE chains at T = 20, two at the metric's own T = 1000, Chikusei);
        # --detail adds the other modes
        pm = ("fp16", "bf16", "fp32") if args.detail else ((args.precision,) if args.precision in ("fp16", "bf16", "fp32") else ("fp16",))
        lm = () if args.no_parity_long else tuple(m for m in pm if m != "bf16")
        parity = chain_parity(dev, modes=pm, long_modes=lm, orth_net=gd.denoise_fn)
        if bf16 is not None and parity.get("bf16"):
            bf16["meets_north_star"] = parity["bf16"]["meets_north_star"]
        if fp32 is not None and parity.get("fp32"):
            fp32["meets_north_star"] = parity["fp32"]["meets_north_star"]
        log('parity done')
    small = None
    if solo and args.detail and not args.no_small and args.precision == HEADLINE and batch > 8 * GROUPS:
        # the same step at the per-GPU share of BASELINE configs[3] on 8 GPUs (8 patches = 40 latents) and at one CAVE image (the
        # reference's own use: 5 latents): what strong scaling and single-image latency are made of
        small = {}
        with torch.no_grad():
            for pp in (8, 1):
                b = pp * GROUPS
                r = gd.make_run(cond[:b].contiguous(), wrap=True)
                for _ in range(n_warm):                         # the wide-set steps of the chain's start, then the fp16 set's eager steps and captures
                    r.step()
                torch.cuda.synchronize()
                n = max(50, min(300, args.steps))
                t0 = time.perf_counter()
                for _ in range(n):
                    r.step()
                torch.cuda.synchronize()
                d = time.perf_counter() - t0
                small["%d_latents" % b] = dict(value=n * b / d, unit="denoise-steps*batch/s", ms_per_step=d / n * 1e3, steps=n,
                                               note="steady-state fp16-mode steps (behind the chain's warm-up)")
                del r
        log('small batches done')
    short = None
    if solo and args.detail and not args.no_small and args.precision == HEADLINE:
        # the reference's SHIPPED chain length (config/sr_sr3_16_128.json:98,104: n_timestep 20 for validation) end to end through the product's
        # entry point: one p_sample_loop_batched call over this GPU's batch = 20 steps, all eight wide-set steps of the policy (1 fp32 + 7 fp32h) among them.
        # The FIRST call of a process pays five eager steps (weight packing) and five graph captures; every later call on the same shapes
        # replays the kept slot (diffusion._GraphSlot) from its first step - what a validation loop over images sees.
        from hsi_dmgasr_amd.sr3_modules import diffusion as _dm
        with torch.no_grad():
            gd20 = _dm.GaussianDiffusion(gd.denoise_fn, image_size=128, channels=3, loss_type="l1", conditional=True).to(dev).eval()
            gd20.set_loss(dev)
            gd20.set_new_noise_schedule(dict(SCHED, n_timestep=20), dev)
            gd20.noise, gd20.seed = "philox", 2
            calls = []
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                x0 = gd20.p_sample_loop_batched(cond)
                torch.cuda.synchronize()
                calls.append((time.perf_counter() - t0) * 1e3)
            assert torch.isfinite(x0).all()
            later = sorted(calls[1:])[1]
            short = dict(chain_steps=20, batch=batch, first_call_ms=calls[0], later_call_ms=later, calls_ms=calls,
                         value_first_call=20 * batch / (calls[0] * 1e-3), value_later_calls=20 * batch / (later * 1e-3),
                         unit="denoise-steps*batch/s", note="whole p_sample_loop_batched calls (20 steps: 1 on the fp32 kernel set, 7 on fp32h, 12 on the dithered fp16 sets); "
                         "later calls replay the first call's captured steps")
            del gd20, x0
        torch.cuda.empty_cache()
        log('short chain done')
    gae_rec = gae_chik = None
    if rank == 0 and args.detail and not args.no_gae:
        with torch.no_grad():
            gae_rec = gae_bench(dev, patches)
            # BASELINE configs[2]: Chikusei, 128 bands, n_subs 16 / n_ovls 4 -> G = 11 groups (AE.py:263-280)
            gae_chik = gae_bench(dev, min(patches, 16), bands=128, n_subs=16, n_ovls=4, groups=11, flops=dict(encode=92.5e9, decode=92.5e9 + 3.8e9))
        log('gae done')
    train_rec = None
    if solo and args.detail and not args.no_train:
        train_rec = train_bench(dev)
        log('train step done')
    cpu = None
    if solo and not args.no_cpu_baseline:
        sd = {k: v.detach().cpu() for k, v in gd.denoise_fn.state_dict().items()}
        del gd
        torch.cuda.empty_cache()
        cpu = cpu_baseline(sd=sd)
        log('cpu baseline done')

    if rank == 0:
        total_batch = total_patches * GROUPS
        # The metric is the 1000-step p_sample_loop: `value` is the rate of a WHOLE chain under the precision policy.  A timed window that
        # holds fewer of the policy's wide-set steps (fp32, fp32h) than their share of a chain (the driver's 20 steps hold none: they are the first eight
        # of a chain and the warm-up consumed them) would read optimistic, so `value` / `ms_per_step` are then the per-chain mix of the
        # measured step times - (T - n) x window step + the n wide-set steps, each set's graph timed right behind the window - and the raw
        # window figures stay in `config`.  A window that is a whole chain (the default 1000 steps) IS the mix and is reported as measured.
        win_value, win_ms = args.steps * total_batch / dt, dt / args.steps * 1e3
        use_mix = value_is_mix(mix, others, args.steps, run_T)
        value, ms_step = (mix["value_chain_mix"], mix["ms_per_step_chain_mix"]) if use_mix else (win_value, win_ms)
        head = {
            "metric": "UNet denoise-steps/sec x batch, CAVE 31-band 16->128, 1000-step p_sample_loop",
            "value": _r(value, 5),
            "unit": "denoise-steps*batch/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": _r(ms_step, 5),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": {"fp16": "fp16", "fp16x1": "fp16", "fp16x2": "fp16", "bf16": "bf16", "fp32": "f32 (bf16x3)"}[args.precision],
            "data": "synthetic (orthogonal-init weights seed 0, N(0,1) latents clipped to +-2.5, Philox noise)",
            "config": {"workload": "SR3 UNet 97.8M (6->3 ch, inner 64, mults 1-2-4-8-8, attn@16) p_sample step on "
                                   "GAE latents 3x128x128, cosine T=1000, BASELINE configs[%d]" % (3 if scaling == "strong" else 1),
                       "patches_per_gpu": patches, "total_patches": total_patches, "groups_per_patch": GROUPS,
                       "batch_per_gpu": batch, "global_batch": total_batch, "parallelism": "dp%d" % world,
                       "precision_mode": DTYPE[args.precision],
                       "warmup_steps_run": n_warm,       # >= --warmup: every kernel set's eager step and capture lie before the clock starts
                       "wide_set_steps_in_window": n_other,
                       "value_is": "per-chain mix of the measured step times" if use_mix else "the timed window as measured",
                       "value_window": _r(win_value, 5), "ms_per_step_window": _r(win_ms, 5),
                       "ms_per_step_chain_mix": None if mix is None else _r(mix["ms_per_step_chain_mix"], 5),
                       "value_chain_mix": None if mix is None else _r(mix["value_chain_mix"], 5),
                       # one step on each of the policy's wide kernel sets (fp32: the first step of a chain; fp32h: the next seven)
                       "ms_per_step_wide_sets": None if mix is None else {o["mode"]: _r(o["ms_per_step"], 5) for o in mix["other_modes"]},
                       "steps_per_chain_wide_sets": None if mix is None else {o["mode"]: o["steps_per_chain"] for o in mix["other_modes"]},
                       # the reference's shipped validation chain length (T = 20): all of the policy's wide-set steps in 20 steps
                       "value_T20": None if mix is None else _r(mix["value_T20"], 5)},
            "rccl_ranks": dist.get_world_size() if use_dist else 1, "allgather_ms": _r(allgather_ms),
            # (the timed window, min / max over the ranks)
            "rank_ms_per_step": {"min": _r(dt_min / args.steps * 1e3, 5), "max": _r(dt / args.steps * 1e3, 5)},
        }
        detail = {"line": None, "chain_mix": mix, "roofline": roof, "parity": parity, "bf16_mode": bf16, "fp32_mode": fp32,
                  "small_batches": small, "short_chain_T20": short, "gae": gae_rec, "gae_chikusei": gae_chik, "train_step": train_rec, "cpu_baseline": cpu}
        dpath = None
        try:
            line = compact_line(head, roof, parity, cpu, args.precision, os.path.relpath(args.detail_out, ROOT))
            detail["line"] = line
            with open(args.detail_out, "w") as f:
                json.dump(detail, f, indent=1)
            dpath = args.detail_out
        except OSError:
            line = compact_line(head, roof, parity, cpu, args.precision, None)       # read-only tree: the line alone
        flush_c_stdio()
        print(json.dumps(line, separators=(",", ":")), flush=True)
    if use_dist:
        dist.barrier()                     # rank 0's extra measurements are done: every rank leaves the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
