"""Oracle: counter-based normal generator (Philox4x32-10 + Box-Muller), numpy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference draws its sampler noise with torch.randn / randn_like
(diffusion.py:174,192), whose device stream cannot be reproduced on another
device.  The build therefore defines its own stateless generator, restated here
bit-for-bit on the integer side (Salmon et al., "Parallel random numbers: as
easy as 1, 2, 3", SC'11; Philox4x32 with 10 rounds) so that the HIP sampler and
this oracle consume identical noise:

  counter = (q & 0xffffffff, q >> 32, stream, 0)   q = element_index // 4
  key     = (seed & 0xffffffff, seed >> 32)
  4 output words -> 4 uniforms u = ((w >> 8) + 0.5) * 2**-24  (fp32, in (0,1))
  normals: r0 = sqrt(-2 ln u0), (z0, z1) = r0 * (cos, sin)(2 pi u1); same for (u2,u3)
  element 4q+j takes z_j.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32; all inputs uint32 arrays (broadcastable)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _W0)
            k1 = np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def normal(seed, stream, n):
    """n fp32 standard normals for (seed, stream); element e uses counter e // 4."""
    nq = (n + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    c0 = (q & _MASK).astype(np.uint32)
    c1 = (q >> np.uint64(32)).astype(np.uint32)
    c2 = np.full(nq, stream, dtype=np.uint32)
    c3 = np.zeros(nq, dtype=np.uint32)
    w = philox4x32(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = [((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24) for x in w]
    two_pi = np.float32(6.283185307179586)
    out = np.empty((nq, 4), dtype=np.float32)
    for p in range(2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[2 * p]))
        th = two_pi * u[2 * p + 1]
        out[:, 2 * p] = r * np.cos(th)
        out[:, 2 * p + 1] = r * np.sin(th)
    return out.reshape(-1)[:n]
