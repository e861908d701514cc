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


def sam_common_support(truth, got, ref):
    """The reference's SAM index (eval_hsi.py:47-65) SKIPS pixels whose predicted spectrum is exactly zero, so it is discontinuous
    where a spectrum sits at the clamp(0, 1) boundary: a pixel that is all-zero in one cube and 1e-5 in one band in the other
    enters / leaves the mean and moves it by (angle_pixel - mean) / N - about 1.4e-3 degrees per pixel on the 128 x 128 fixtures,
    whatever the size of the deviation that flipped it.  Returns (number of pixels whose membership differs between `got` and
    `ref`, |SAM(truth, got) - SAM(truth, ref)| in degrees over the pixels BOTH cubes keep).  All arrays (H, W, C)."""
    t = truth.astype(np.float32).reshape(-1, truth.shape[2])
    a = got.astype(np.float32).reshape(-1, got.shape[2])
    b = ref.astype(np.float32).reshape(-1, ref.shape[2])
    nt, na, nb = (np.linalg.norm(v, axis=1) for v in (t, a, b))
    flips = int(np.count_nonzero((na != 0) != (nb != 0)))
    ok = (nt != 0) & (na != 0) & (nb != 0)

    def sam(p, npn):
        return float(np.sum(np.arccos(np.sum(t[ok] * p[ok], axis=1) / (nt[ok] * npn[ok]))) / np.count_nonzero(ok) * 180.0 / np.pi)
    return flips, abs(sam(a, na) - sam(b, nb))
