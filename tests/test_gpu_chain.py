"""GPU parity at the shipped configuration and over the shipped chain length, and at the batch sizes the benchmark's
kernel dispatch needs.

  * chain.npz: the reference's validation iteration (sr_gae.py:436-474) on the 97.8 M-parameter UNet, its T = 20 cosine
    chain, one CAVE image (5 spectral groups, pretrained CAVE autoencoder), run by the imported reference with the noise
    of tests/golden/synth.py:chain_noise (tests/golden/make_golden_chain.py);
  * north-star gates ("1e-3 relative; PSNR/SAM within 0.01 dB / 0.001"): the fp32 mode and the fp16 mode (a POLICY: dithered one-pass
    fp16 kernel sets on the low-gain steps of a chain, the fp32 kernel set on its high-gain steps - precision.py) are held to ALL of
    them, the latents also on the support the reference did not clamp; the bf16 set meets the PSNR bound only and is gated at its
    measured deviation with head-room (DSAM_MAX / LATENT_MAX below say so explicitly);
  * the SAM bound is applied to the reference's STRICT index (eval_hsi.py:47-65); the common-support index replaces it only under the
    two conditions of helpers.sam_gate (at most two flipped pixels, each at the clamp boundary in both cubes), a strict miss is an
    xfail of its own (test_strict_sam_index_of_the_headline_mode), and a CONTINUOUS companion - SAM on the un-clamped decoded cubes -
    is gated on every chain;
  * the benchmark's dispatch (256-cout items on 8 waves, multi-round persistent loops, XCD tile remap) only engages at
    B >= 36: one forward of the shipped UNet at B = 40 against the oracle on the host, with the launch set asserted.
"""
import numpy as np
import pytest
import torch

from gpu_util import check, fill_synth, log_err
from helpers import (chain_fixture, jload, load_npz, reference_unclamped_cube, rel_err, rel_err_unsaturated, sam_common_support,
                     sam_continuous, sam_gate, synth_tensor)
from synth import CHAIN_CHIKUSEI, CHAIN_LONG_SET, CHAIN_SET, CHAIN_TUNED_ON

pytestmark = pytest.mark.gpu

FULL = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8], attn_res=[16],
            res_blocks=2, image_size=128)
# north_star: "Outputs match the reference PyTorch CPU path within 1e-3 relative fp32 (PSNR/SAM within 0.01 dB / 0.001)".
#   fp32  : meets every bound with four orders of magnitude to spare (bf16 hi + lo operands, three MFMA passes).
#   fp16  : meets every continuous bound - 11-bit significands, the weight rounding (the one error that is the same in every step
#           of the chain: a bias, not noise) dithered over the steps, the high-gain steps on the fp32 kernel set;
#           tests/precision_emul.py reproduces the device numbers on the host and shows which rounding contributes what.  The
#           reference's SAM index is met strictly on ten of the eleven chains; on orth:4:20 ONE pixel sits at the index's
#           discontinuity (see below and test_strict_sam_index_of_the_headline_mode).
#   bf16  : meets the PSNR bound (0.0012 dB) but neither the latent bound (7.7e-3) nor the SAM bound (0.012 degrees): 8-bit
#           significands on every MFMA operand and stored activation.  Its gate is the measured value with 2.5x head-room - a
#           regression gate, NOT a north-star claim (bench.py's `parity` object says meets_north_star: false for it).
# fp16x1 (no second weight pass) sits ON the bounds (emulation: 9.9e-4 / 1.0e-3 degrees) and is only logged.
# A note on dSAM on this fixture: the UNet's weights are synthetic, so the decoded cube is not an image - 55 % of it is clamped
# to 0, SAM is 56 degrees, and the index (mean angle over pixels whose spectra are non-zero, eval_hsi.py:47-65; 67 all-zero pixels, 113 with one band left) moves by
# 1.4e-3 degrees whenever ONE pixel's spectrum crosses between all-zero and not.  dSAM therefore counts such crossings rather
# than measuring an angle: a mode passes when its perturbation (latents ~5e-4) flips none.  The CONTINUOUS companion - the same angle
# on the un-clamped decoded cubes (ours: the product's decoder on our latents; the reference's: the oracle's decoder, pinned to the
# reference at 1e-6, on the fixture's latents) - has no support to flip and is gated at the same 0.001 degrees on every chain.
NORTH_STAR = dict(latents=1e-3, dpsnr=0.01, dsam=0.001)
DSAM_CONT_MAX = {"fp32": 0.001, "fp16": 0.001, "fp16x2": 0.001, "fp16x1": 0.01, "bf16": 0.03}
DPSNR_MAX = {"fp32": 0.01, "fp16": 0.01, "fp16x2": 0.01, "fp16x1": 0.01, "bf16": 0.01}
DSAM_MAX = {"fp32": 0.001, "fp16": 0.001, "fp16x2": 0.001, "fp16x1": 0.01, "bf16": 0.03}
LATENT_MAX = {"fp32": 1e-3, "fp16": 1e-3, "fp16x2": 1e-3, "fp16x1": 2e-3, "bf16": 2e-2}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _full_unet(dev, prec):
    from hsi_dmgasr_amd.sr3_modules import unet
    u = unet.UNet(dropout=0.2, precision=prec, **FULL).to(dev).eval()
    sd = fill_synth(u, "unet_full.")
    return u, sd


def _run_chain(dev, prec, fixture):
    """One member of the chain fixture set through the product's per-image driver: returns the measured deviations."""
    from hsi_dmgasr_amd import gae, pipeline
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import metrics
    weights, draw, steps = fixture
    g, sd, hr, sr, cn = chain_fixture(*fixture)
    u = unet.UNet(dropout=0.2, precision=prec, **FULL).to(dev).eval()
    u.load_state_dict(sd)
    gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=steps, linear_start=1e-6, linear_end=1e-2), dev)
    # the autoencoder runs in its fp32 mode in every case: it is 0.02 % of the work and not what the 16-bit gates are about
    m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=8, n_ovls=2, n_colors=31, n_feats=64, precision="fp32").to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in load_npz("gae_cav_state.npz").items()})
    ngr = g["x0"].shape[0]
    x_T = G(np.concatenate([cn(gi, 0) for gi in range(ngr)]), dev)                               # [G,3,H,W]
    noise = G(np.stack([np.concatenate([cn(gi, k) for gi in range(ngr)]) for k in range(1, steps)]), dev)
    y, lat = pipeline.super_resolve(m, gd, G(sr, dev), x_T=x_T, noise=noise, precision=prec)
    y_raw = m.decode_batched(lat, 31)                   # the same decode before sr_gae.py:473-474 clamps it (super_resolve clamps in place)
    torch.cuda.synchronize()
    lat = lat[0].cpu().numpy()
    y = y.cpu().numpy()
    e_lat, e_y = rel_err(lat, g["x0"]), rel_err(y, g["y"])
    e_unsat, sat_frac = rel_err_unsaturated(lat, g["x0"])
    a = hr[0].transpose(1, 2, 0)
    ref = g["y"][0].transpose(1, 2, 0)
    got = y[0].transpose(1, 2, 0)
    dpsnr = abs(metrics.mpsnr(a, got) - metrics.mpsnr(a, ref))
    dsam = abs(metrics.sam_degrees(a, got) - metrics.sam_degrees(a, ref))
    flips, dsam_common, flip_norm = sam_common_support(a, got, ref, detail=True)
    ref_raw = reference_unclamped_cube("%s:%d:%d" % fixture, "gae_cav_state.npz", 31, g["x0"], 8, 2)
    assert rel_err(np.clip(ref_raw, 0.0, 1.0), g["y"][0]) < 2e-5                  # the oracle's decode IS the reference's cube before the clamp
    dsam_cont = abs(sam_continuous(a, y_raw[0].cpu().numpy().transpose(1, 2, 0)) - sam_continuous(a, ref_raw.transpose(1, 2, 0)))
    # the fixture's own indices were computed by the reference's eval_hsi.py: the oracle's restatements must agree on them
    assert abs(metrics.sam_degrees(a, ref) - float(g["sam"])) < 2e-3
    assert abs(metrics.mpsnr(a, ref) - float(g["mpsnr_formula"])) < 1e-4
    assert np.isfinite(lat).all() and np.isfinite(y).all()
    gated, strict = sam_gate(dsam, flips, dsam_common, flip_norm)
    rec = {"cube_rel_err": e_y, "dPSNR_dB": dpsnr, "dSAM_deg": dsam, "fixture": "%s:%d:%d" % fixture,
           "latents_rel_err_unsaturated": e_unsat, "dSAM_unclamped_deg": dsam_cont,
           "psnr_of_ours_vs_reference_cube_dB": metrics.mpsnr(ref, got),         # the reference's decoded cube as the "truth"
           "max_abs_latent_diff": float(np.abs(lat - g["x0"]).max()),
           "clamped_latent_fraction": sat_frac,
           "zero_spectrum_crossings": flips, "dSAM_common_support_deg": dsam_common, "flipped_pixel_norm_over_cube_rms": flip_norm,
           "sam_gate_is_strict_index": strict,
           "meets_north_star": bool(e_lat <= NORTH_STAR["latents"] and e_y <= NORTH_STAR["latents"] and dpsnr <= NORTH_STAR["dpsnr"] and
                                    dsam <= NORTH_STAR["dsam"])}
    log_err("chain_T%d_full_latents" % steps, prec, e_lat, rec)
    LAST_CHAIN_RECORD.clear()
    LAST_CHAIN_RECORD.update(rec, latents_rel_err=e_lat, precision=prec)          # (tools/chain_probe.py prints it)
    CHAIN_RECORDS[(prec, fixture)] = dict(LAST_CHAIN_RECORD)
    # the latents are gated on BOTH supports (all elements; the elements the reference did not clamp), the continuous SAM next to the index
    assert prec not in DSAM_CONT_MAX or dsam_cont <= DSAM_CONT_MAX[prec], (prec, fixture, dsam_cont)
    return max(e_lat, e_unsat), e_y, dpsnr, gated


LAST_CHAIN_RECORD = {}
CHAIN_RECORDS = {}          # (precision, fixture) -> record, for the strict-index test below


@pytest.mark.parametrize("fixture", CHAIN_SET, ids=lambda f: "%s-n%d-T%d" % f)
@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_full_size_T20_chain_against_the_reference_run(dev, prec, fixture):
    """The reference's validation iteration on every T = 20 member of the fixture set: two weight sets (synthetic Gaussian, and the
    reference's own orthogonal initialisation - the weights bench.py times) x three noise / cube draws each.  Round 5's precision
    policy (dithered one-pass fp16 weights, eight fp32-set steps per chain) was selected by EMULATION on the first four
    (tests/precision_emul.py); orth:4:20 and synth:5:20 were generated after it was fixed, like both 1000-step chains and the Chikusei
    chain they are hold-outs (synth.CHAIN_HOLDOUT).  Every mode that claims north_star is held to it on ALL of them."""
    e_lat, e_y, dpsnr, dsam = _run_chain(dev, prec, fixture)
    assert e_lat < LATENT_MAX[prec] and e_y < LATENT_MAX[prec], (prec, fixture, e_lat, e_y)
    assert dpsnr <= DPSNR_MAX[prec] and dsam <= DSAM_MAX[prec], (prec, fixture, dpsnr, dsam, e_lat, e_y)


@pytest.mark.parametrize("fixture", CHAIN_LONG_SET, ids=lambda f: "%s-n%d-T%d" % f)
@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_full_size_T1000_chain_against_the_reference_run(dev, prec, fixture):
    """BASELINE.json's metric is the 1000-step p_sample_loop (reference diffusion.py:177-201): one CAVE image through all 1000
    steps of the cosine chain against the run of the imported reference (tests/golden/chains/*_T1000.npz, make_golden_chains.py;
    999 stored noise draws per group) - with the reference's own orthogonal initialisation (the weights bench.py times) and, a second
    image and draw, with the synthetic weight set."""
    e_lat, e_y, dpsnr, dsam = _run_chain(dev, prec, fixture)
    assert e_lat < LATENT_MAX[prec] and e_y < LATENT_MAX[prec], (prec, e_lat, e_y)
    assert dpsnr <= DPSNR_MAX[prec] and dsam <= DSAM_MAX[prec], (prec, dpsnr, dsam, e_lat, e_y)


def test_strict_sam_index_of_the_headline_mode(dev):
    """The reference's OWN SAM index (eval_hsi.py:47-65, strict) on every CAVE chain in the headline mode - the quantity north_star
    names.  Where the bound is missed, the miss must be the index's discontinuity and nothing else - at most helpers.SAM_MAX_FLIPS
    pixels whose zero-spectrum membership differs, each with a spectrum norm below helpers.SAM_FLIP_NORM x the cube's rms in both
    cubes, the index over the common support and the un-clamped (continuous) index inside 0.001 deg - and is then reported as an
    XFAIL naming the fixture, never absorbed: `pytest -rx` shows it.  (Round 5 / 6 builds: orth:4:20, one pixel whose largest band is
    1.5e-4 in the reference's cube; strict 1.42e-3 deg, 1.1e-4 on the common support.)"""
    from helpers import SAM_FLIP_NORM, SAM_MAX_FLIPS
    misses = []
    for fixture in tuple(CHAIN_SET) + tuple(CHAIN_LONG_SET):
        rec = CHAIN_RECORDS.get(("fp16", fixture))
        if rec is None:
            _run_chain(dev, "fp16", fixture)
            rec = CHAIN_RECORDS[("fp16", fixture)]
        if rec["dSAM_deg"] <= NORTH_STAR["dsam"]:
            continue
        # a strict miss: only the discontinuity may explain it
        assert 0 < rec["zero_spectrum_crossings"] <= SAM_MAX_FLIPS, rec
        assert rec["flipped_pixel_norm_over_cube_rms"] <= SAM_FLIP_NORM, rec
        assert rec["dSAM_common_support_deg"] <= NORTH_STAR["dsam"] and rec["dSAM_unclamped_deg"] <= NORTH_STAR["dsam"], rec
        misses.append("%s: strict dSAM %.2e deg (%d pixel(s) at the clamp boundary, norm %.1e x cube rms; common support %.1e deg, un-clamped "
                      "%.1e deg)" % (rec["fixture"], rec["dSAM_deg"], rec["zero_spectrum_crossings"], rec["flipped_pixel_norm_over_cube_rms"],
                                     rec["dSAM_common_support_deg"], rec["dSAM_unclamped_deg"]))
    log_err("chain_strict_sam_misses", "fp16", float(len(misses)), {"misses": misses})
    if misses:
        pytest.xfail("north_star's strict SAM bound (0.001 deg) missed at the index's discontinuity: " + "; ".join(misses))


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_chikusei_chain_against_the_reference_run(dev, prec):
    """BASELINE configs[2]: Chikusei, 128 bands -> n_subs 16 / n_ovls 4 -> ELEVEN group latents through the 20-step chain on the shipped
    UNet (the reference's orthogonal initialisation), encoded and decoded by the reference's PRETRAINED Chikusei autoencoder
    (GAE_pretrained/GAE_4_Chi.pth, stored as data in tests/golden/gae_chi_state.npz), against the run of the imported reference
    (tests/golden/chains/chi_orth_n3_T20.npz: the eleven latents, every fourth band of the decoded cube, the indices of the whole cube)."""
    from hsi_dmgasr_amd import gae, pipeline
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import metrics
    from synth import chain_cubes_draw, chain_noise_draw
    weights, draw, steps = CHAIN_CHIKUSEI
    g = load_npz("chains/chi_%s_n%d_T%d.npz" % CHAIN_CHIKUSEI)
    _, sd, _, _, _ = chain_fixture("orth", 0, 20)              # the orthogonal weights (rebuilt once per session, checked against the probes)
    u = unet.UNet(dropout=0.2, precision=prec, **FULL).to(dev).eval()
    u.load_state_dict(sd)
    gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=steps, linear_start=1e-6, linear_end=1e-2), dev)
    m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=16, n_ovls=4, n_colors=128, n_feats=64, precision="fp32").to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in load_npz("gae_chi_state.npz").items()})
    assert m.G == 11 == int(g["groups"][0])
    hr, sr = chain_cubes_draw(draw, 128)
    ngr = g["x0"].shape[0]
    x_T = G(np.concatenate([chain_noise_draw(draw, gi, 0) for gi in range(ngr)]), dev)
    noise = G(np.stack([np.concatenate([chain_noise_draw(draw, gi, k) for gi in range(ngr)]) for k in range(1, steps)]), dev)
    y, lat = pipeline.super_resolve(m, gd, G(sr, dev), x_T=x_T, noise=noise, precision=prec)
    y_raw = m.decode_batched(lat, 128)                  # before the clamp of sr_gae.py:473-474
    torch.cuda.synchronize()
    lat, y = lat[0].cpu().numpy(), y.cpu().numpy()
    e_lat, e_y = rel_err(lat, g["x0"]), rel_err(y[:, ::4], g["y_sub4"])
    e_unsat, sat_frac = rel_err_unsaturated(lat, g["x0"])
    a, got = hr[0].transpose(1, 2, 0), y[0].transpose(1, 2, 0)
    dpsnr = abs(metrics.mpsnr(a, got) - float(g["mpsnr_formula"]))
    dsam = abs(metrics.sam_degrees(a, got) - float(g["sam_oracle"]))
    assert abs(float(g["sam_oracle"]) - float(g["sam"])) < 2e-3                        # the restatement against the reference's eval_hsi on its cube
    # the continuous companion of the index (un-clamped cubes; the fixture stores a quarter of the reference's CLAMPED cube: the oracle's
    # decode of the reference's latents is checked against it first)
    ref_raw = reference_unclamped_cube("chi:%s:%d:%d" % CHAIN_CHIKUSEI, "gae_chi_state.npz", 128, g["x0"], 16, 4)
    assert rel_err(np.clip(ref_raw[::4], 0.0, 1.0), g["y_sub4"][0]) < 2e-5
    dsam_cont = abs(sam_continuous(a, y_raw[0].cpu().numpy().transpose(1, 2, 0)) - sam_continuous(a, ref_raw.transpose(1, 2, 0)))
    log_err("chain_chikusei_T20_latents", prec, e_lat, {"cube_rel_err_every_4th_band": e_y, "dPSNR_dB": dpsnr, "dSAM_deg": dsam,
                                                        "latents_rel_err_unsaturated": e_unsat, "clamped_latent_fraction": sat_frac,
                                                        "dSAM_unclamped_deg": dsam_cont, "fixture": "chi:%s:%d:%d" % CHAIN_CHIKUSEI})
    assert np.isfinite(lat).all() and np.isfinite(y).all()
    assert max(e_lat, e_unsat) < LATENT_MAX[prec] and e_y < LATENT_MAX[prec], (prec, e_lat, e_unsat, e_y)
    assert dpsnr <= DPSNR_MAX[prec] and dsam <= DSAM_MAX[prec] and dsam_cont <= DSAM_CONT_MAX[prec], (prec, dpsnr, dsam, dsam_cont)


@pytest.mark.parametrize("prec", ["fp16x2", "fp16x1"])
def test_T20_chain_ab_forms_of_the_fp16_mode(dev, prec):
    """The A/B forms (second weight pass everywhere / nowhere) on the tuned-on fixture: logged, gated loosely."""
    e_lat, e_y, dpsnr, dsam = _run_chain(dev, prec, CHAIN_TUNED_ON)
    assert e_lat < LATENT_MAX[prec] and e_y < LATENT_MAX[prec], (e_lat, e_y)
    assert dpsnr <= DPSNR_MAX[prec] and dsam <= DSAM_MAX[prec], (prec, dpsnr, dsam, e_lat, e_y)


def test_weight_dither_is_what_lets_one_weight_pass_through(dev, monkeypatch):
    """The policy's claim in one comparison on a reference chain (orth:0:20, the weights bench.py times): with the SAME eight fp32-set
    steps, one-pass fp16 weights that are plainly rounded in every step ("fp16x1") deviate more than one-pass weights dithered over the
    steps (the "fp16" policy: measured 6.5e-4 against 5.4e-4 on this chain; 7.7e-4 against 3.0e-4 on the 1000-step chains,
    profiles/r05_ab/chain_probe_T1000.txt) - the difference is the weight rounding's bias."""
    e_dith = _run_chain(dev, "fp16", ("orth", 0, 20))[0]
    e_plain = _run_chain(dev, "fp16x1", ("orth", 0, 20))[0]
    log_err("chain_T20_dither_vs_plain_rounding", "fp16", e_dith, {"plain_rounding": e_plain, "fixture": "orth:0:20"})
    assert e_dith < 0.9 * e_plain and e_dith < NORTH_STAR["latents"], (e_dith, e_plain)


def test_benchmark_batch_chain_against_the_reference_run(dev):
    """The BENCHMARK's kernel dispatch - 240 latents per GPU: 256-cout items on 8 waves, multi-round persistent loops (30 rounds per
    launch on the 128x128 level), XCD tile remap, the four dithered weight sets - against the reference: the orthogonal-weights T = 20 chain (five group
    latents by the imported reference) replicated 48 times into one batch of 240 in the headline mode; EVERY copy is held to north_star's
    1e-3 on the latents, and the copies agree with each other (the path has no cross-sample arithmetic)."""
    from hsi_dmgasr_amd import ops
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    fixture = ("orth", 0, 20)
    g, sd, hr, sr, cn = chain_fixture(*fixture)
    steps, copies = fixture[2], 48
    u = unet.UNet(dropout=0.2, precision="fp16", **FULL).to(dev).eval()
    u.load_state_dict(sd)
    gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=steps, linear_start=1e-6, linear_end=1e-2), dev)
    ngr = g["x0"].shape[0]
    rep = lambda a: G(a, dev).repeat(copies, 1, 1, 1)                                     # copy c = rows c*ngr .. c*ngr + ngr - 1
    z = rep(g["z"])
    x_T = rep(np.concatenate([cn(gi, 0) for gi in range(ngr)]))
    noise = torch.stack([rep(np.concatenate([cn(gi, k) for gi in range(ngr)])) for k in range(1, steps)])
    recs = []
    ops.set_conv_probe(recs)
    try:
        run = gd.make_run(z, x_T=x_T, noise=noise, precision="fp16")
        with torch.no_grad():
            n_hi = sum(1 for m in run.modes if m in ("fp32", "fp32h"))
            assert n_hi == 8 and run.modes[:2] == ["fp32", "fp32h"] and run.modes[n_hi + 1] == "fp16d1"
            for _ in range(n_hi + 1):                                                     # the eight wide-set steps and the first fp16-set one
                run.step()
            torch.cuda.synchronize()
            del recs[:]
            gd.use_graph, keep = False, gd.use_graph                                      # one eager fp16-set step under the probe: which kernels the batch takes
            try:
                run.step()
            finally:
                gd.use_graph = keep
            labels = sorted({r["kernel"] for r in recs})
            for _ in range(steps - n_hi - 2):
                run.step()
            torch.cuda.synchronize()
            assert run.steps_done == steps
    finally:
        ops.set_conv_probe(None)
    for need in ("conv_v2 bn256 8x16", "conv_v3 bn64", "up4", "dn4", "conv1x1_g"):
        assert any(need in l for l in labels), (need, labels)
    assert not any(l.startswith("conv_igemm") for l in labels), labels
    lat = run.x.cpu().numpy().reshape(copies, ngr, 3, 128, 128)
    errs = [rel_err(lat[c], g["x0"]) for c in range(copies)]
    spread = max(rel_err(lat[c], lat[0]) for c in range(1, copies))
    log_err("chain_T20_batch240_worst_copy", "fp16", max(errs), {"best_copy": min(errs), "copy_to_copy": spread, "fixture": "%s:%d:%d" % fixture})
    assert max(errs) < NORTH_STAR["latents"], (max(errs), min(errs))
    assert spread < 1e-6, spread


def test_benchmark_dispatch_forward_matches_the_oracle_at_batch_40(dev):
    """The launch set of the benchmark (bench.py at >= 8 patches per GPU) against the CPU oracle: shipped UNet, bf16 mode,
    B = 40, one forward; the probe labels prove which kernels ran."""
    from hsi_dmgasr_amd import ops
    from oracle import sr3_unet
    u, sd = _full_unet(dev, "bf16")
    B = 40
    x = synth_tensor("dispatch40.x", (B, 6, 128, 128))
    gam = np.linspace(0.02, 0.98, B, dtype=np.float32).reshape(B, 1)
    recs = []
    ops.set_conv_probe(recs)
    try:
        y = u(G(x, dev), G(gam, dev))
        torch.cuda.synchronize()
    finally:
        ops.set_conv_probe(None)
    labels = sorted({r["kernel"] for r in recs})
    # (the 8x8 level's two-image 256-cout items need >= 128 work items = batch 128: next test)
    for need in ("conv_v2 bn256 8x16", "conv_v3 bn64", "up4", "dn4", "conv1x1_g"):
        assert any(need in l for l in labels), (need, labels)
    assert any(l.startswith("conv1x1_g") and l.endswith(" gn") for l in labels), labels           # attention qkv with the GN prologue
    assert not any(l.startswith("conv_igemm") for l in labels), labels                             # nothing on the generic kernel
    with torch.no_grad():
        want = sr3_unet.unet_forward(sd, FULL, torch.from_numpy(x), torch.from_numpy(gam))
    e = check("unet_full_b40_bench_dispatch", "bf16", y, want, tol=2e-2)
    per = [rel_err(y[i].cpu().numpy(), want[i].numpy()) for i in range(B)]
    log_err("unet_full_b40_worst_sample", "bf16", max(per))
    assert max(per) < 3e-2, max(per)


def test_8x8_level_two_image_256_cout_items_match_the_oracle(dev):
    """The 8x8 level's benchmark form (two-image 8x8 tiles, 256-cout items on 8 waves) engages from 128 work items: a
    512 -> 512 ResnetBlock on 8x8 maps at B = 132 against the oracle block."""
    from hsi_dmgasr_amd import ops
    from hsi_dmgasr_amd.sr3_modules import unet
    from oracle import sr3_unet
    m = unet.ResnetBlock(512, 512, noise_level_emb_dim=64, norm_groups=32).to(dev).eval()
    m.precision = "bf16"
    sd = fill_synth(m, "res8x8.")
    B = 132
    x = synth_tensor("res8x8.x", (B, 512, 8, 8))
    t = synth_tensor("res8x8.t", (B, 1, 64))
    recs = []
    ops.set_conv_probe(recs)
    try:
        y = m(G(x, dev), G(t, dev))
        torch.cuda.synchronize()
    finally:
        ops.set_conv_probe(None)
    labels = sorted({r["kernel"] for r in recs})
    assert any("conv_v2 bn256 8x8x2" in l for l in labels), labels
    with torch.no_grad():
        want = sr3_unet.resnet_block(sd, "", torch.from_numpy(x), torch.from_numpy(t), 32)
    check("resblock_8x8_b132_bn256", "bf16", y, want, tol=1e-2)


def test_graph_replayed_philox_chain_at_batch_40_with_wrap(dev):
    """8-step Philox chain at B = 40 on the tiny UNet with wrap=True (the benchmark's run object: eager first step, graph
    capture, replays; the device-side counter restarts after t = 0) against the oracle chain."""
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    from oracle import diffusion as odiff, sr3_unet
    cfg = jload(load_npz("unets.npz")["tiny.cfg_json"])
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16, precision="fp32").to(dev).eval()
    sd = fill_synth(u, "unet_tiny.")
    opt = dict(schedule="cosine", n_timestep=8, linear_start=1e-6, linear_end=1e-2)
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(opt, dev)
    gd.noise, gd.seed = "philox", 314
    B = 40
    cond = G(synth_tensor("wrap40.cond", (B, 3, 16, 16)), dev)
    run = gd.make_run(cond, wrap=True)
    with torch.no_grad():
        for _ in range(8):
            run.step()
        torch.cuda.synchronize()
        first = run.x.clone()
        assert int(run.t_ptr.item()) == 7                       # wrapped
        assert run.graph is not None                            # steps 3.. were graph replays
    sched = odiff.noise_schedule(opt)
    den = lambda x, gam: sr3_unet.unet_forward(sd, cfg, x, gam)
    nf = odiff.philox_noise_fn(314, (B, 3, 16, 16))
    xo = nf(8)
    for i in reversed(range(8)):
        xo = odiff.p_sample_step(den, sched, xo, cond.cpu(), i, nf(i) if i > 0 else None)
    check("philox_chain_b40_wrap", "fp32", first, xo, tol=1e-3)
    # a second lap continues from x_0 of the first with the same per-step noise streams
    with torch.no_grad():
        for _ in range(8):
            run.step()
        torch.cuda.synchronize()
    for i in reversed(range(8)):
        xo = odiff.p_sample_step(den, sched, xo, cond.cpu(), i, nf(i) if i > 0 else None)
    check("philox_chain_b40_wrap_lap2", "fp32", run.x, xo, tol=1e-3)
    # without wrap the run refuses to step past the end of the chain
    run2 = gd.make_run(cond[:2])
    with torch.no_grad():
        run2.run_all()
        with pytest.raises(RuntimeError):
            run2.step()


def test_precision_schedule_runs_the_high_gain_steps_on_the_fp32_kernels(dev):
    """precision.step_precision in the reverse loop: a chain in the "fp16" policy runs its first step (update gain 31.6 on the cosine
    schedule) on the fp32 kernel set - after it its state is BIT-IDENTICAL to a chain run in the fp32 mode on the same Philox noise -
    the next seven (1.5, 0.83, 0.58, 0.45, 0.37, 0.31, 0.27) on the fp32h set (fp32 storage, fp16 operands: the states differ by the
    operands' rounding, ~1e-4) and the rest on the fp16 kernels with the weight dither of phase (step % 4); one captured graph per kernel
    set, all in one memory pool; the schedule is a property of the chain's position, so it repeats after a wrap."""
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16, precision="fp16").to(dev).eval()
    fill_synth(u, "unet_tiny.")
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
    gd.set_loss(dev)
    T = 20
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=T, linear_start=1e-6, linear_end=1e-2), dev)
    gd.noise, gd.seed = "philox", 77
    cond = G(synth_tensor("sched.cond", (3, 3, 16, 16)), dev)
    a = gd.make_run(cond, wrap=True)                            # the module's mode: the "fp16" policy
    b = gd.make_run(cond, wrap=True, precision="fp32")
    assert a.modes == ["fp32"] + ["fp32h"] * 7 + ["fp16d%d" % (k % 4) for k in range(8, T)] and b.modes == ["fp32"] * T
    with torch.no_grad():
        a.step(); b.step()
        torch.cuda.synchronize()
        assert torch.equal(a.x, b.x)
        for _ in range(7):
            a.step(); b.step()
        torch.cuda.synchronize()
        e8 = rel_err(a.x.cpu().numpy(), b.x.cpu().numpy())
        assert not torch.equal(a.x, b.x) and e8 < 3e-4, e8
        for _ in range(T - 8):
            a.step(); b.step()
        torch.cuda.synchronize()
        assert not torch.equal(a.x, b.x) and rel_err(a.x.cpu().numpy(), b.x.cpu().numpy()) < 2e-3
        # (the fp32 set has run ONE step so far - eagerly; its capture is the first step of the second lap)
        assert set(a.graphs) == {"fp16d0", "fp16d1", "fp16d2", "fp16d3", "fp32h"} and a.graph is a.graphs["fp16d3"] and int(a.t_ptr.item()) == T - 1      # wrapped
        for _ in range(8):                                      # second lap: the eight high-gain steps again on the fp32 kernels (graph replays)
            a.step(); b.step()
        torch.cuda.synchronize()
        assert "fp32" in a.graphs
    assert torch.isfinite(a.x).all() and a.steps_done == T + 8


def _tiny_gd(dev, T, prec="fp16", seed_prefix="unet_tiny."):
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16, precision=prec).to(dev).eval()
    fill_synth(u, seed_prefix)
    gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
    gd.set_loss(dev)
    gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=T, linear_start=1e-6, linear_end=1e-2), dev)
    return gd


def test_captured_steps_are_reused_by_the_next_p_sample_loop_call(dev):
    """The reference's validation loop calls p_sample_loop once per image and group (sr_gae.py:458-465); a fresh ReverseRun pays six
    eager steps (fp32, fp32h, four dither phases) and five graph captures (the fp32 set runs one step per chain: captured by the second
    call), more than half of the shipped 20-step chain.  GaussianDiffusion keeps
    the finished call's slot (static buffers + graphs + pool): the next call on the same shapes copies its inputs in and replays from
    its FIRST step - same graphs, results bit-identical to an eager (graph-free) run of the same inputs; a call with other inputs in
    between does not leak into it; changed weights drop the slot; the results handed out are copies, not the slot's buffers."""
    T, B = 20, 3
    gd = _tiny_gd(dev, T)
    eager = _tiny_gd(dev, T)
    eager.use_graph = False
    cond = [G(synth_tensor("slot.cond%d" % i, (B, 3, 16, 16)), dev) for i in range(2)]
    x_T = [G(synth_tensor("slot.xT%d" % i, (B, 3, 16, 16)), dev) for i in range(2)]
    noise = [G(synth_tensor("slot.noise%d" % i, (T - 1, B, 3, 16, 16)), dev) for i in range(2)]
    want = [eager.p_sample_loop_batched(cond[i], x_T=x_T[i], noise=noise[i]) for i in range(2)]
    a0 = gd.p_sample_loop_batched(cond[0], x_T=x_T[0], noise=noise[0])
    assert len(gd._graph_cache) == 1 and not gd._graph_cache[0].busy
    slot = gd._graph_cache[0]
    graphs = dict(slot.graphs)
    assert set(graphs) == {"fp32h", "fp16d0", "fp16d1", "fp16d2", "fp16d3"}
    keep = a0.clone()
    b1 = gd.p_sample_loop_batched(cond[1], x_T=x_T[1], noise=noise[1])        # other inputs through the SAME slot: replays only
    a2 = gd.p_sample_loop_batched(cond[0], x_T=x_T[0], noise=noise[0])
    torch.cuda.synchronize()
    assert len(gd._graph_cache) == 1 and gd._graph_cache[0] is slot and all(slot.graphs[k] is g for k, g in graphs.items())
    assert set(slot.graphs) == set(graphs) | {"fp32"}                           # (the chain's one fp32-set step: captured by the second call)
    assert torch.equal(a0, keep) and a0.data_ptr() != slot.x.data_ptr()         # the first call's result was a copy: the later calls did not touch it
    assert torch.equal(a0, want[0]) and torch.equal(b1, want[1]) and torch.equal(a2, want[0])
    # the stock entry points go through the same slots (continous: the snapshots are copied out too)
    gd.noise, gd.seed = eager.noise, eager.seed = "philox", 11
    r1 = gd.p_sample_loop(cond[0], continous=True)
    r2 = gd.p_sample_loop(cond[0], continous=True)
    torch.cuda.synchronize()
    assert torch.equal(r1, r2) and torch.equal(r1, eager.p_sample_loop(cond[0], continous=True)) and len(gd._graph_cache) == 2
    # new weights: the slots' graphs address the old packed weights and are dropped, the next call captures afresh
    other = _tiny_gd(dev, T, seed_prefix="unet_tiny_b.")
    gd.denoise_fn.load_state_dict(other.denoise_fn.state_dict())
    other.use_graph = False
    c = gd.p_sample_loop_batched(cond[0], x_T=x_T[0], noise=noise[0])
    torch.cuda.synchronize()
    assert slot not in gd._graph_cache and torch.equal(c, other.p_sample_loop_batched(cond[0], x_T=x_T[0], noise=noise[0]))
    assert not torch.equal(c, a0)
    # a sampler switch re-points the coefficient tables: no slot survives it
    gd.set_sampler("ddim", steps=10)
    assert gd._graph_cache == []


def test_graphs_of_one_pool_replay_in_another_order_than_they_were_captured(dev):
    """The kernel sets of a chain (fp32, fp32h, four dither phases) are captured into ONE memory pool: each replay may overwrite what the
    others left behind, which is safe only while nothing allocated inside a capture outlives it except through the static buffers.
    A 22-step chain with wrap captures in the order fp32h, d0, d1, d2, d3, then fp32 (the second lap's first step) and replays ... d0,
    d1 | fp32, fp32h x 7, d0 ... - d1 followed by fp32, an order no capture saw; three laps against the same chain run eagerly
    (graph-free) on the same Philox noise must be bit-identical."""
    T, B = 22, 4
    gd, eager = _tiny_gd(dev, T), _tiny_gd(dev, T)
    eager.use_graph = False
    for g in (gd, eager):
        g.noise, g.seed = "philox", 23
    cond = G(synth_tensor("pool.cond", (B, 3, 16, 16)), dev)
    a, b = gd.make_run(cond, wrap=True), eager.make_run(cond, wrap=True)
    assert a.modes[8:] == ["fp16d%d" % (k % 4) for k in range(8, T)] and a.modes[-1] == "fp16d1"
    with torch.no_grad():
        for lap in range(3):
            for _ in range(T):
                a.step(); b.step()
            torch.cuda.synchronize()
            assert torch.equal(a.x, b.x), lap
    assert len(a.graphs) == 6 and not b.graphs


def test_sharded_driver_under_an_rccl_group_of_one(dev):
    """pipeline.super_resolve_sharded with the default process group initialised on RCCL ("nccl", world size 1): same
    cubes as super_resolve."""
    import os
    import torch.distributed as dist
    from hsi_dmgasr_amd import gae, parallel, pipeline
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                      image_size=16, precision="fp32").to(dev).eval()
        fill_synth(u, "unet_tiny.")
        gd = diffusion.GaussianDiffusion(u, image_size=16, channels=3, conditional=True)
        gd.set_loss(dev)
        gd.set_new_noise_schedule(dict(schedule="cosine", n_timestep=4, linear_start=1e-6, linear_end=1e-2), dev)
        gd.noise, gd.seed = "philox", 5
        m = gae.GAE(gae.Encoder, gae.Decoder, n_subs=8, n_ovls=2, n_colors=31, n_feats=64, precision="fp32").to(dev).eval()
        fill_synth(m, "gae_cave_synth.")
        parallel.broadcast_module_(gd, src=0)
        parallel.broadcast_module_(m, src=0)
        cubes = G(np.abs(synth_tensor("sharded.x", (3, 31, 16, 16), scale=0.3)), dev)
        a = pipeline.super_resolve_sharded(m, gd, cubes)
        b, _ = pipeline.super_resolve(m, gd, cubes)
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                      # the group really is RCCL
        torch.cuda.synchronize()
        assert dist.get_backend() == "nccl" and float(t[0]) == 1.0
        assert a.shape == b.shape and torch.equal(a, b)
    finally:
        if created:
            dist.destroy_process_group()


def test_repacking_follows_in_place_reinitialisation(dev):
    """forward -> init_weights_orthogonal(seed=1) -> forward must change (the packed-weight cache keys on
    Parameter._version, which writes through `.data` would not bump), and a broadcast after a warm-up forward too."""
    from hsi_dmgasr_amd.init import init_weights_orthogonal
    from hsi_dmgasr_amd.sr3_modules import unet
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=32, channel_mults=[1, 2], attn_res=[8], res_blocks=1,
                  image_size=16, precision="bf16").to(dev).eval()
    init_weights_orthogonal(u, seed=0)
    x = G(synth_tensor("repack.x", (2, 6, 16, 16)), dev)
    gam = G(np.array([[0.5], [0.2]], dtype=np.float32), dev)
    a = u(x, gam).clone()
    init_weights_orthogonal(u, seed=1)
    b = u(x, gam).clone()
    init_weights_orthogonal(u, seed=0)
    c = u(x, gam).clone()
    torch.cuda.synchronize()
    assert not torch.equal(a, b)
    assert torch.equal(a, c)


def _run_bench(tmp_path, *flags):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    side = str(tmp_path / "bench_detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--detail-out", side] + list(flags),
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    return last, json.loads(last), json.load(open(side))


@pytest.mark.gpu
def test_bench_line_is_the_compact_contract_object(tmp_path):
    """`python bench.py --steps 20 --warmup 5` as the driver runs it (at 12 patches = 60 latents): the LAST stdout line is ONE strict
    JSON object under 4 KB with the contract's keys, `roofline` (dominant kernel instance, the conv_v3 family as one row, the whole
    step), `cpu_baseline` (both cases) and `parity` (headline mode, worst case over the T = 20 reference chains incl. Chikusei);
    the full objects are in the side file, not on stdout."""
    last, d, side = _run_bench(tmp_path, "--patches", "12", "--steps", "20", "--warmup", "5")
    assert len(last.encode()) <= 4096, len(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["dtype"] == "fp16"
    cfg = d["config"]
    assert cfg["batch_per_gpu"] == 60 and "workload" in cfg and "model" not in cfg
    assert abs(d["value"] - 20 * 60 / (d["ms_per_step"] * 20e-3)) < 1e-3 * d["value"]
    # the policy's eight fp32-set steps of a chain sit in the warm-up of a 20-step window: the line says so, and `value` IS the per-chain
    # mix (992 window steps + 8 fp32-set steps per 1000), not the optimistic window rate, which stays in config; the same policy on the
    # reference's shipped 20-step chain is reported beside it
    assert cfg["wide_set_steps_in_window"] == 0 and cfg["value_is"].startswith("per-chain mix")
    assert d["value"] == cfg["value_chain_mix"] and d["ms_per_step"] == cfg["ms_per_step_chain_mix"]
    wide = cfg["ms_per_step_wide_sets"]
    assert cfg["steps_per_chain_wide_sets"] == {"fp32": 1, "fp32h": 7}
    assert cfg["value_window"] > d["value"] and cfg["ms_per_step_window"] < d["ms_per_step"] < wide["fp32h"] < wide["fp32"]
    assert 0.3 * d["value"] < cfg["value_T20"] < d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and 0.05 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["launches"] >= 1 and rf["avg_launch_us"] > 0 and rf["algorithmic_flops_per_launch"] > 0
    assert abs(rf["achieved"] - rf["algorithmic_flops_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e12) < 1e-2 * rf["achieved"]
    assert 0.05 < rf["whole_step"]["frac"] < rf["frac"] + 0.2 and rf["conv_v3_family"]["launches"] == 10
    assert rf["fused_resnetblock_hbm_frac"] < rf["resnetblock_launch_hbm_frac"]
    par = d["parity"]
    # (orth:4:20 - a hold-out - sits at the SAM index's discontinuity: one pixel's zero-spectrum membership differs, see helpers.sam_gate)
    assert par["mode"] == "fp16" and par["meets_north_star_with_sam_on_common_support"] is True and par["n_fixtures"] == 11
    assert par["meets_north_star"] is (par["dSAM_deg"] <= 1e-3) and par["meets_north_star"] is (par["strict_sam_misses"] == [])
    assert par["latents_rel_err"] <= par["latents_rel_err_unsaturated"] < 1e-3 and par["dSAM_unclamped_deg"] <= 1e-3
    assert any("T1000" in k for k in par["worst_of"])
    assert par["latents_rel_err"] < 1e-3 and par["cube_rel_err"] < 1e-3 and par["dSAM_deg_on_common_support"] <= 1e-3 and par["dPSNR_dB"] <= 0.01
    assert d["meets_north_star"] is par["meets_north_star"]
    assert d["rank_ms_per_step"]["min"] <= d["rank_ms_per_step"]["max"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and 0 < cb["value"] < d["value"] and set(cb["cases"]) == {"batch_1", "batch_5"}
    assert cb["value"] == cb["cases"]["batch_1"]["value"]
    # side file: the same line + the full objects
    assert side["line"] == d and len(side["roofline"]["kernels"]) >= 8 and len(side["parity"]["fp16"]["fixtures"]) == par["n_fixtures"]
    assert side["bf16_mode"] is None and side["gae"] is None and side["train_step"] is None        # --detail legs not run


@pytest.mark.gpu
def test_bench_detail_legs(tmp_path):
    """`python bench.py --detail` at a reduced size: the secondary legs land in the side file - the other precision modes (bf16 does
    not meet north_star and is labelled so), small batches, the group autoencoder (CAVE and Chikusei) on the HBM axis, the training
    step - and the stdout line stays the compact one."""
    last, d, side = _run_bench(tmp_path, "--patches", "12", "--steps", "6", "--warmup", "2", "--detail", "--no-cpu-baseline")
    assert len(last.encode()) <= 4096 and "cpu_baseline" not in d
    par = side["parity"]
    assert par["fp16"]["meets_north_star_with_sam_on_common_support"] is True and par["fp32"]["meets_north_star"] is True and par["bf16"]["meets_north_star"] is False
    assert any("T1000" in k for k in par["fp16"]["fixtures"]) and not any("T1000" in k for k in par["bf16"]["fixtures"])
    assert side["bf16_mode"]["value"] > 0 and side["bf16_mode"]["meets_north_star"] is False
    assert side["fp32_mode"]["value"] > 0 and side["fp32_mode"]["roofline"]["mfma_passes_per_product"] == 3
    assert set(side["small_batches"]) == {"40_latents", "5_latents"} and all(v["value"] > 0 for v in side["small_batches"].values())
    # the shipped 20-step chain through p_sample_loop_batched: later calls replay the first call's captured steps
    sc = side["short_chain_T20"]
    assert sc["chain_steps"] == 20 and 0 < sc["later_call_ms"] < sc["first_call_ms"] and sc["value_later_calls"] > sc["value_first_call"]
    for key in ("gae", "gae_chikusei"):
        for mode in ("fp32", "fp16"):
            assert side[key][mode]["encode_ms"] > 0 and "dPSNR_dB_vs_fp32_mode" in side[key][mode]
            assert 0 < side[key][mode]["encode_hbm"]["unit_frac_of_hbm"] < side[key][mode]["encode_hbm"]["launch_frac_of_hbm"] < 1
        assert side[key]["fp16"]["within_0.01dB_0.001deg"] is True
    assert side["gae_chikusei"]["cube"] == "128x128x128, G=11"
    assert side["train_step"]["bf16"]["graph_ms_per_step"] > 0
