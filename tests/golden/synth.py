"""Deterministic synthetic parameters keyed by state_dict name.

Used by tests/golden/make_golden.py (to fill the *reference* modules before
capturing their outputs) and by the tests (to rebuild the very same weights
without storing them).  numpy PCG64 seeded with crc32(key) ^ seed, so the values
do not depend on torch's RNG or on module construction order.
"""
import zlib

import numpy as np


def synth_param(key, shape, seed=0):
    rng = np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if len(shape) >= 2:                       # conv / linear weight: ~unit gain
        fan_in = int(np.prod(shape[1:]))
        w = rng.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
    elif leaf == "weight":                    # GroupNorm gamma
        w = 1.0 + 0.1 * rng.standard_normal(shape)
    else:                                     # biases / GroupNorm beta
        w = 0.05 * rng.standard_normal(shape)
    return w.astype(np.float32)


def synth_state_dict(shapes, seed=0):
    """shapes: {key: shape}.  Returns {key: np.float32 array}."""
    return {k: synth_param(k, s, seed) for k, s in shapes.items()}


def synth_tensor(tag, shape, seed=0, scale=1.0):
    rng = np.random.Generator(np.random.PCG64((zlib.crc32(tag.encode()) ^ (seed * 0x85EBCA6B)) & 0xFFFFFFFF))
    return (scale * rng.standard_normal(tuple(shape))).astype(np.float32)
