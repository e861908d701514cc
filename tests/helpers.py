"""Shared helpers for the test-suite (fixtures loading, synthetic weights)."""
import json
import os

import numpy as np
import torch

from synth import (AUGMENT_SHAPE, COLOR_CASES, IMRESIZE_CASES, METRIC_CASES, augment_input, imresize_input, metric_pair,
                   synth_param, synth_tensor)  # noqa: F401  tests/golden/synth.py (path added by conftest)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth_sd(shapes, prefix="", seed=0):
    """{key: shape} -> {key: torch fp32 tensor}; the generator key is prefix+key."""
    return {k: torch.from_numpy(synth_param(prefix + k, s, seed)) for k, s in shapes.items()}


def sub_shapes(all_shapes, prefix):
    """Select the keys of one op from ops.npz's shapes_json and strip the prefix."""
    return {k[len(prefix):]: tuple(v) for k, v in all_shapes.items() if k.startswith(prefix)}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def jload(arr):
    return json.loads(str(arr))


FULL_UNET = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8], attn_res=[16],
                 res_blocks=2, image_size=128)
_ORTH = {}


def chain_fixture(weights, draw, steps):
    """One member of the chain fixture set (tests/golden/make_golden_chain.py, make_golden_chains.py): returns
    (g, sd, hr, sr, noise) with g the stored reference outputs, sd the UNet's state dict as torch tensors (rebuilt, and for the
    orthogonal set CHECKED against the fixture's probes), (hr, sr) the cubes and noise(group, k) the draws."""
    from oracle import sr3_unet
    from synth import chain_cubes_draw, chain_noise_draw, chain_weights_check, orth_state_dict
    g = load_npz("chain.npz" if (weights, draw, steps) == ("synth", 0, 20) else "chains/%s_n%d_T%d.npz" % (weights, draw, steps))
    shapes = sr3_unet.unet_param_shapes(FULL_UNET)
    if weights == "synth":
        sd = synth_sd(shapes, "unet_full.")
    else:
        if "sd" not in _ORTH:
            keys = [str(k) for k in g["w_keys"]]
            _ORTH["sd"] = orth_state_dict(keys, shapes)
            chain_weights_check(_ORTH["sd"], keys, g["w_probe"])
        sd = _ORTH["sd"]
    hr, sr = chain_cubes_draw(draw)
    return g, sd, hr, sr, (lambda gi, k: chain_noise_draw(draw, gi, k))


def sam_common_support(truth, got, ref, detail=False):
    """The reference's SAM index (eval_hsi.py:47-65) SKIPS pixels whose predicted spectrum is exactly zero, so it is discontinuous
    where a spectrum sits at the clamp(0, 1) boundary: a pixel that is all-zero in one cube and 1e-5 in one band in the other
    enters / leaves the mean and moves it by (angle_pixel - mean) / N - about 1.4e-3 degrees per pixel on the 128 x 128 fixtures,
    whatever the size of the deviation that flipped it.  Returns (number of pixels whose membership differs between `got` and
    `ref`, |SAM(truth, got) - SAM(truth, ref)| in degrees over the pixels BOTH cubes keep); detail=True adds the largest spectrum
    norm of a flipped pixel in EITHER cube relative to the reference cube's rms (0.0 without flips): sam_gate() only accepts the
    common-support index when that is tiny, i.e. when every flipped pixel really sits at the clamp boundary.  All arrays (H, W, C)."""
    t = truth.astype(np.float32).reshape(-1, truth.shape[2])
    a = got.astype(np.float32).reshape(-1, got.shape[2])
    b = ref.astype(np.float32).reshape(-1, ref.shape[2])
    nt, na, nb = (np.linalg.norm(v, axis=1) for v in (t, a, b))
    flip = (na != 0) != (nb != 0)
    flips = int(np.count_nonzero(flip))
    ok = (nt != 0) & (na != 0) & (nb != 0)

    def sam(p, npn):
        return float(np.sum(np.arccos(np.sum(t[ok] * p[ok], axis=1) / (nt[ok] * npn[ok]))) / np.count_nonzero(ok) * 180.0 / np.pi)
    d = abs(sam(a, na) - sam(b, nb))
    if not detail:
        return flips, d
    rms = float(np.sqrt(np.mean(b.astype(np.float64) ** 2)))
    worst = float(max(na[flip].max(), nb[flip].max()) / rms) if flips else 0.0
    return flips, d, worst


# The SAM bound is applied to the reference's own (strict) index unless the two cubes disagree about which pixels the index skips,
# and then to the common-support index ONLY IF the disagreement is the index's discontinuity and nothing else: at most SAM_MAX_FLIPS
# pixels, each with a spectrum norm of at most SAM_FLIP_NORM x the cube's rms in BOTH cubes (a pixel that left the support with a
# real spectrum is a deviation, not a boundary effect, and keeps the strict index as the gated quantity).
SAM_MAX_FLIPS = 2
SAM_FLIP_NORM = 1e-3


def sam_gate(dsam, flips, dsam_common, flip_norm_rel):
    """(value the SAM bound is applied to, is it the strict index?) - see the comment above."""
    if flips == 0 or flips > SAM_MAX_FLIPS or flip_norm_rel > SAM_FLIP_NORM:
        return dsam, True
    return dsam_common, False


def sam_continuous(truth, pred):
    """Mean spectral angle (degrees) over ALL pixels with a non-zero true spectrum, for UN-CLAMPED decoded cubes: without the
    clamp(0, 1) of sr_gae.py:473-474 no predicted spectrum is exactly zero, the index has no support to flip and is a continuous
    function of the cube - the companion of the reference's index on fixtures whose decoded cubes are half zeros.  (H, W, C)."""
    t = truth.astype(np.float64).reshape(-1, truth.shape[2])
    p = pred.astype(np.float64).reshape(-1, pred.shape[2])
    nt, npn = np.linalg.norm(t, axis=1), np.linalg.norm(p, axis=1)
    ok = (nt != 0) & (npn != 0)
    cos = np.clip(np.sum(t[ok] * p[ok], axis=1) / (nt[ok] * npn[ok]), -1.0, 1.0)
    return float(np.mean(np.arccos(cos)) * 180.0 / np.pi)


def rel_err_unsaturated(a, b):
    """Relative deviation over the elements the reference did NOT clamp (|b| < 1; diffusion.py:164 clamps x_0 to [-1, 1]): a
    saturated element carries no error and all of its weight in the norm - 38 % of the reference's T = 20 latents are exactly +-1,
    which flatters the plain relative error by ~1.5x.  Returns (error on the unsaturated support, fraction of saturated elements)."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    m = np.abs(b) < 1.0
    return float(np.linalg.norm((a - b)[m]) / max(np.linalg.norm(b[m]), 1e-30)), float(1.0 - m.mean())


_UNCLAMPED = {}


def reference_unclamped_cube(key, gae_state_npz, n_colors, x0, n_subs, n_ovls):
    """The ORACLE's decode (oracle/gae.py, pinned to the reference at 1e-6) of a fixture's reference latents WITHOUT the final clamp:
    (C, H, W) float32, cached per fixture."""
    if key not in _UNCLAMPED:
        from oracle import gae as ogae
        gsd = {k: torch.from_numpy(v) for k, v in load_npz(gae_state_npz).items()}
        with torch.no_grad():
            z = torch.from_numpy(np.ascontiguousarray(x0))
            _UNCLAMPED[key] = ogae.gae_decode(gsd, n_colors, [z[i:i + 1] for i in range(z.shape[0])], n_subs, n_ovls)[0].numpy()
    return _UNCLAMPED[key]
