// Philox4x32-10 counter-based generator (Salmon et al., SC'11) + Box-Muller; restated bit-for-bit in oracle/philox.py.
#pragma once
#include "common.h"

namespace hsidm {

// ---- Philox4x32-10 (Salmon et al., SC'11) + Box-Muller; restated bit-for-bit in oracle/philox.py ----
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ void philox_normal4(uint64_t q, uint32_t stream_id, uint64_t seed, float (&z)[4]) {
    const Philox4 r = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float k = 5.9604644775390625e-08f;                         // 2^-24
    const float u0 = ((float)(r.x >> 8) + 0.5f) * k, u1 = ((float)(r.y >> 8) + 0.5f) * k;
    const float u2 = ((float)(r.z >> 8) + 0.5f) * k, u3 = ((float)(r.w >> 8) + 0.5f) * k;
    const float two_pi = 6.283185307179586f;
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    const float t0 = two_pi * u1, t1 = two_pi * u3;
    z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0);
    z[2] = r1 * cosf(t1); z[3] = r1 * sinf(t1);
}

}  // namespace hsidm
