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
