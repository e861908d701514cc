// Reverse-diffusion update with a counter-based normal generator, NCHW<->NHWC boundary conversions,
// and the group-autoencoder's elementwise pieces.  All HBM-streaming kernels (16-B vectors per lane).
#include "common.h"
#include "philox.h"
#include "../../include/hsidm.h"

namespace hsidm {

__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint32_t stream_id) {
    const int64_t nq = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        float z[4];
        philox_normal4((uint64_t)q, stream_id, seed, z);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * q + j < n) out[4 * q + j] = z[j];
    }
}

// x_{t-1} = c1*clamp(a*x - b*eps) + c2*x + [t>0]*z*exp(0.5*logvar)     (diffusion.py:142-175)
__global__ __launch_bounds__(256) void p_sample_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                              const float* __restrict__ coef, const int32_t* __restrict__ t_ptr,
                                                              int T, const float* __restrict__ noise, int64_t noise_stride, uint64_t seed,
                                                              int64_t n, float* __restrict__ snap, int32_t inter) {
    const int t = *t_ptr;
    if (t < 0 || t >= T) return;            // a replay past the end of the chain (the host wrapper refuses it too) must not index the tables
    const float a = coef[t * 5 + 0], b = coef[t * 5 + 1], c1 = coef[t * 5 + 2], c2 = coef[t * 5 + 3];
    const float sigma = expf(0.5f * coef[t * 5 + 4]);
    const float* zsrc = noise ? noise + (int64_t)(T - 1 - t) * noise_stride : nullptr;
    float* sdst = nullptr;
    if (snap && (t % inter) == 0) {
        // snapshots are taken at t = k*inter, k = (T-1)/inter .. 0  ->  slot index counts from the first
        const int first = (T - 1) / inter;
        sdst = snap + (int64_t)(first - t / inter) * n;
    }
    const int64_t nq = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            if (zsrc) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * q + j < n) z[j] = zsrc[4 * q + j];
            } else {
                philox_normal4((uint64_t)q, (uint32_t)t, seed, z);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = 4 * q + j;
            if (i < n) {
                const float xv = x[i];
                float x0 = a * xv - b * eps[i];
                x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                float m = c1 * x0 + c2 * xv;
                if (t > 0) m = m + z[j] * sigma;
                x[i] = m;
                if (sdst) sdst[i] = m;
            }
        }
    }
}

__global__ void step_advance_kernel(int32_t* t_ptr, int32_t wrap_T) {
    int32_t t = *t_ptr - 1;
    if (t < 0 && wrap_T > 0) t = wrap_T - 1;
    *t_ptr = t;
}


// ---- forward process and training objective (diffusion.py:213-250) -----------------------------------
// x_noisy = gamma_b * x0 + sqrt(1 - gamma_b^2) * noise, gamma per sample
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const float* __restrict__ gamma, float* __restrict__ out,
                                                       int64_t per_sample, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float g = gamma[i / per_sample];
        out[i] = g * x0[i] + sqrtf(1.0f - g * g) * noise[i];
    }
}

constexpr int kLossBlocks = 256;
// stage 1: block b sums its fixed strided subset (fp32 per thread, fp64 across the block) -> partial[b]
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                           int kind, double* __restrict__ partial) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float d = a[i] - b[i];
        acc += kind == 0 ? fabsf(d) : d * d;
    }
    __shared__ double red[256];
    red[threadIdx.x] = (double)acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// stage 2: fixed-order sum of the partials -> out[0] (fp32, like the reference's scalar loss)
__global__ __launch_bounds__(64) void loss_final_kernel(const double* __restrict__ partial, int nb, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += partial[i];
        out[0] = (float)s;
    }
}

// ---- NCHW fp32 planes -> NHWC (storage type), with channel concat / slicing / zero padding -----------
template <typename ActT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ s0, const int64_t* __restrict__ off0, int C0,
                                                           const float* __restrict__ s1, const int64_t* __restrict__ off1, int C1,
                                                           ActT* __restrict__ out, int HW, int Cpad) {
    const int e = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* a = s0 + off0[e] + p;
    const float* b = s1 ? s1 + off1[e] + p : nullptr;
    ActT* o = out + ((size_t)e * HW + p) * Cpad;
    for (int c8 = 0; c8 < Cpad; c8 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c8 + k;
            v[k] = c < C0 ? a[(size_t)c * HW] : (c < C0 + C1 ? b[(size_t)(c - C0) * HW] : 0.f);
        }
        Vec8<ActT>::store(o + c8, v);
    }
}

// NHWC (storage type) -> NCHW fp32.  32 pixels x 64 channels per workgroup through LDS so that both the
// loads (channels contiguous) and the stores (pixels contiguous) are coalesced.
template <typename ActT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const ActT* __restrict__ src, float* __restrict__ out, int HW, int C) {
    __shared__ float tile[32][65];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    for (int i = t; i < 32 * 64; i += 256) {
        const int pp = i >> 6, cc = i & 63;
        float v = 0.f;
        if (p0 + pp < HW && c0 + cc < C) v = to_f32<ActT>(src[((size_t)b * HW + p0 + pp) * C + c0 + cc]);
        tile[pp][cc] = v;
    }
    __syncthreads();
    for (int i = t; i < 32 * 64; i += 256) {
        const int cc = i >> 5, pp = i & 31;
        if (p0 + pp < HW && c0 + cc < C) out[((size_t)b * C + c0 + cc) * HW + p0 + pp] = tile[pp][cc];
    }
}

// ---- group autoencoder elementwise pieces ------------------------------------------------------------------
// out = res_scale * r * ca[b][c] + skip (+ skip2), NHWC, 8 channels per thread
template <typename ActT>
__global__ __launch_bounds__(256) void ca_apply_kernel(const ActT* __restrict__ r, const float* __restrict__ ca,
                                                       const ActT* __restrict__ skip, const ActT* __restrict__ skip2,
                                                       float res_scale, ActT* __restrict__ out, int HW, int C, int64_t nvec) {
    const int nv = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % nv);
        const int64_t pix = i / nv;
        const int b = (int)(pix / HW);
        float a[8], s[8], s2[8];
        Vec8<ActT>::load(r + i * 8, a);
        Vec8<ActT>::load(skip + i * 8, s);
        if (skip2) Vec8<ActT>::load(skip2 + i * 8, s2);
        const float* cab = ca + (size_t)b * C + cv * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = res_scale * (a[k] * cab[k]) + s[k];
            if (skip2) v += s2[k];
            a[k] = v;
        }
        Vec8<ActT>::store(out + i * 8, a);
    }
}

// out = (1 + gb[b][c]) * x + gb[b][C + c]: FeatureWiseAffine with use_affine_level (reference unet.py:44-47), NHWC, 8 channels per thread
template <typename ActT>
__global__ __launch_bounds__(256) void film_affine_kernel(const ActT* __restrict__ x, const float* __restrict__ gb, ActT* __restrict__ out,
                                                          int HW, int C, int64_t nvec) {
    const int nv = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % nv);
        const int b = (int)(i / nv / HW);
        float a[8];
        Vec8<ActT>::load(x + i * 8, a);
        const float* g = gb + (size_t)b * 2 * C + cv * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fmaf(1.0f + g[k], a[k], g[C + k]);
        Vec8<ActT>::store(out + i * 8, a);
    }
}

// y[b][c][p] = mean over the groups g covering band c of dec[b*G+g][c-start[g]][p]   (AE.py:286-295)
__global__ __launch_bounds__(256) void overlap_average_kernel(const float* __restrict__ dec, const int32_t* __restrict__ start,
                                                              int G, int n_subs, int C, int HW, float* __restrict__ y) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    float acc = 0.f;
    float cnt = 0.f;
    for (int g = 0; g < G; ++g) {
        const int s = start[g];
        if (c >= s && c < s + n_subs) {
            acc += dec[(((size_t)b * G + g) * n_subs + (c - s)) * HW + p];
            cnt += 1.f;
        }
    }
    y[((size_t)b * C + c) * HW + p] = acc / cnt;
}

}  // namespace hsidm

using namespace hsidm;

static inline int grid_for(int64_t work_items, int per_block = 256, int cap = 256 * 8) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

extern "C" int hsidm_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t stream_id, void* stream) {
    if (!out || n <= 0) return HSIDM_E_BADARG;
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, seed, stream_id);
    return (int)hipGetLastError();
}

extern "C" int hsidm_p_sample_update(float* x, const float* eps, const float* coef, const int32_t* t_ptr, int T,
                                     const float* noise, int64_t noise_stride, uint64_t seed, int64_t n, float* snap,
                                     int32_t inter, void* stream) {
    if (!x || !eps || !coef || !t_ptr || T <= 0 || n <= 0 || (snap && inter <= 0)) return HSIDM_E_BADARG;
    hipLaunchKernelGGL(p_sample_update_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, eps, coef,
                       t_ptr, T, noise, noise_stride, seed, n, snap, inter);
    return (int)hipGetLastError();
}

extern "C" int hsidm_step_advance(int32_t* t_ptr, int32_t wrap_T, void* stream) {
    if (!t_ptr) return HSIDM_E_BADARG;
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t_ptr, wrap_T);
    return (int)hipGetLastError();
}


extern "C" int hsidm_q_sample(const float* x0, const float* noise, const float* gamma, float* out, int B,
                              int64_t per_sample, void* stream) {
    if (!x0 || !noise || !gamma || !out || B <= 0 || per_sample <= 0) return HSIDM_E_BADARG;
    int64_t n = (int64_t)B * per_sample;
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x0, noise, gamma, out, per_sample, n);
    return (int)hipGetLastError();
}

extern "C" int hsidm_loss_workspace_bytes(void) { return kLossBlocks * (int)sizeof(double); }

extern "C" int hsidm_loss_sum(const float* a, const float* b, int64_t n, int kind, void* workspace, float* out, void* stream) {
    if (!a || !b || !workspace || !out || n <= 0 || (kind != HSIDM_LOSS_L1 && kind != HSIDM_LOSS_L2)) return HSIDM_E_BADARG;
    int nb = grid_for(n, 256, kLossBlocks);
    hipLaunchKernelGGL(loss_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, n, kind, (double*)workspace);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, nb, out);
    return (int)hipGetLastError();
}

extern "C" int hsidm_nchw_to_nhwc(int prec, const float* src0, const int64_t* off0, int C0, const float* src1,
                                  const int64_t* off1, int C1, void* out, int Bout, int HW, int Cpad, void* stream) {
    if (!src0 || !off0 || !out || C0 <= 0 || C1 < 0 || (C1 > 0 && !(src1 && off1)) || (Cpad & 7) || Cpad < C0 + C1 || Bout <= 0 || HW <= 0)
        return HSIDM_E_BADARG;
    dim3 grid((HW + 255) / 256, Bout);
    hipStream_t s = (hipStream_t)stream;
    const float* s1 = C1 > 0 ? src1 : nullptr;
    if (prec == HSIDM_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, grid, dim3(256), 0, s, src0, off0, C0, s1, off1, C1, (bf16*)out, HW, Cpad);
    else if (prec == HSIDM_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, grid, dim3(256), 0, s, src0, off0, C0, s1, off1, C1, (f16*)out, HW, Cpad);
    else if (prec == HSIDM_F32X3)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, s, src0, off0, C0, s1, off1, C1, (float*)out, HW, Cpad);
    else
        return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_nhwc_to_nchw(int prec, const void* src, float* out, int B, int HW, int C, void* stream) {
    if (!src || !out || B <= 0 || HW <= 0 || C <= 0) return HSIDM_E_BADARG;
    dim3 grid((HW + 31) / 32, (C + 63) / 64, B);
    hipStream_t s = (hipStream_t)stream;
    if (prec == HSIDM_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src, out, HW, C);
    else if (prec == HSIDM_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16>, grid, dim3(256), 0, s, (const f16*)src, out, HW, C);
    else if (prec == HSIDM_F32X3)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, s, (const float*)src, out, HW, C);
    else
        return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_ca_apply(int prec, const void* r, const float* ca, const void* skip, const void* skip2, float res_scale,
                              void* out, int B, int HW, int C, void* stream) {
    if (!r || !ca || !skip || !out || (C & 7) || B <= 0 || HW <= 0) return HSIDM_E_BADARG;
    const int64_t nvec = (int64_t)B * HW * (C >> 3);
    hipStream_t s = (hipStream_t)stream;
    if (prec == HSIDM_BF16)
        hipLaunchKernelGGL(ca_apply_kernel<bf16>, dim3(grid_for(nvec)), dim3(256), 0, s, (const bf16*)r, ca, (const bf16*)skip,
                           (const bf16*)skip2, res_scale, (bf16*)out, HW, C, nvec);
    else if (prec == HSIDM_F16)
        hipLaunchKernelGGL(ca_apply_kernel<f16>, dim3(grid_for(nvec)), dim3(256), 0, s, (const f16*)r, ca, (const f16*)skip,
                           (const f16*)skip2, res_scale, (f16*)out, HW, C, nvec);
    else if (prec == HSIDM_F32X3)
        hipLaunchKernelGGL(ca_apply_kernel<float>, dim3(grid_for(nvec)), dim3(256), 0, s, (const float*)r, ca, (const float*)skip,
                           (const float*)skip2, res_scale, (float*)out, HW, C, nvec);
    else
        return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_film_affine(int prec, const void* x, const float* gamma_beta, void* out, int B, int HW, int C, void* stream) {
    if (!x || !gamma_beta || !out || B <= 0 || HW <= 0 || C <= 0 || (C & 7)) return HSIDM_E_BADARG;
    const int64_t nvec = (int64_t)B * HW * (C >> 3);
    hipStream_t s = (hipStream_t)stream;
    if (prec == HSIDM_BF16)
        hipLaunchKernelGGL(film_affine_kernel<bf16>, dim3(grid_for(nvec)), dim3(256), 0, s, (const bf16*)x, gamma_beta, (bf16*)out, HW, C, nvec);
    else if (prec == HSIDM_F16)
        hipLaunchKernelGGL(film_affine_kernel<f16>, dim3(grid_for(nvec)), dim3(256), 0, s, (const f16*)x, gamma_beta, (f16*)out, HW, C, nvec);
    else if (prec == HSIDM_F32X3)
        hipLaunchKernelGGL(film_affine_kernel<float>, dim3(grid_for(nvec)), dim3(256), 0, s, (const float*)x, gamma_beta, (float*)out, HW, C, nvec);
    else
        return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_overlap_average(const float* dec, const int32_t* start, int G, int n_subs, int B, int C, int HW,
                                     float* y, void* stream) {
    if (!dec || !start || !y || G <= 0 || n_subs <= 0 || B <= 0 || C <= 0 || HW <= 0) return HSIDM_E_BADARG;
    dim3 grid((HW + 255) / 256, C, B);
    hipLaunchKernelGGL(overlap_average_kernel, grid, dim3(256), 0, (hipStream_t)stream, dec, start, G, n_subs, C, HW, y);
    return (int)hipGetLastError();
}
