// GroupNorm statistics, noise-level embedding / FiLM projections, channel attention vector.
// All HBM-bound or tiny kernels: coalesced 16-B channel vectors, wave-shuffle / LDS reductions,
// fp32 arithmetic throughout.
#include "common.h"
#include "../../include/hsidm.h"

namespace hsidm {

// ---------------------------------------------------------------------------------------------------
// gn_partial: grid (nsplit, B).  Each workgroup streams a contiguous pixel range of one image,
// thread t owns channel vector (t % nvec) of pixel rows t / nvec, t / nvec + rows, ...; per-channel
// partial sums are combined across the rows through LDS in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------
template <typename ActT>
__global__ __launch_bounds__(256) void gn_partial_kernel(const ActT* __restrict__ s0, const ActT* __restrict__ s1,
                                                         int C0, int C1, int HW, int nsplit,
                                                         float2* __restrict__ part) {
    __shared__ float red[256 * 8 * 2];
    const int C = C0 + C1;
    const int nvec = C >> 3;
    const int rows = 256 / nvec;
    const int split = blockIdx.x, b = blockIdx.y;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p_begin = split * per;
    const int p_end = min(HW, p_begin + per);
    const int t = threadIdx.x;
    const int cvi = t % nvec, prow = t / nvec;
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = q[k] = 0.f;
    if (prow < rows) {
        const int c = cvi * 8;
        const ActT* src;
        int cs, cl;
        if (c < C0) { src = s0; cs = C0; cl = c; } else { src = s1; cs = C1; cl = c - C0; }
        for (int p = p_begin + prow; p < p_end; p += rows) {
            float v[8];
            Vec8<ActT>::load(src + ((size_t)b * HW + p) * cs + cl, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += v[k]; q[k] = fmaf(v[k], v[k], q[k]); }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[(prow * C + c + k) * 2] = s[k];
            red[(prow * C + c + k) * 2 + 1] = q[k];
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float a = 0.f, d = 0.f;
        for (int r = 0; r < rows; ++r) { a += red[(r * C + c) * 2]; d += red[(r * C + c) * 2 + 1]; }
        part[((size_t)b * nsplit + split) * C + c] = make_float2(a, d);
    }
}

// gn_finalize: grid (groups / GPB, B).  part0 [B][nsplit0][C0] (+ part1 [B][nsplit1][C1] for a channel concat)
//              -> gn_ab [B][C0+C1] = (rstd*gamma, beta - mean*rstd*gamma)
// A workgroup owns GPB consecutive groups (= CW consecutive channels) of one image.  The slab of a 128x128 map has
// 256 partial entries per channel, so the entries are spread over all 256 threads (thread = (channel, row); row r sums
// entries r, r+R, ...) and combined through LDS in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float2* __restrict__ part0, int nsplit0, int C0,
                                                          const float2* __restrict__ part1, int nsplit1, int C1,
                                                          int HW, int groups, int gpb, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, double inv_n,
                                                          float2* __restrict__ ab) {
    __shared__ float scr[512];
    __shared__ float cs[256], cq[256], gm[256], gr[256];
    const int C = C0 + C1;
    const int cpg = C / groups;
    const int cw = gpb * cpg;                      // channels of this workgroup (<= 256)
    const int c0 = blockIdx.x * cw;
    const int b = blockIdx.y, t = threadIdx.x;
    const int R = 256 / cw;
    const int cl = t % cw, r = t / cw;
    float a = 0.f, d = 0.f;
    if (r < R) {
        const int c = c0 + cl;
        // this thread's entries s = r, r + R, ...: four requested at a time, added in order (one load per trip was one exposed
        // round trip per entry - the whole duration of a launch this small)
        const float2* src = c < C0 ? part0 + (size_t)b * nsplit0 * C0 + c : part1 + (size_t)b * nsplit1 * C1 + (c - C0);
        const int ns = c < C0 ? nsplit0 : nsplit1, cs = c < C0 ? C0 : C1;
        int s = r;
        for (; s + 3 * R < ns; s += 4 * R) {
            float2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = src[(size_t)(s + j * R) * cs];
#pragma unroll
            for (int j = 0; j < 4; ++j) { a += v[j].x; d += v[j].y; }
        }
        for (; s < ns; s += R) {
            const float2 v = src[(size_t)s * cs];
            a += v.x;
            d += v.y;
        }
    }
    scr[2 * t] = a;
    scr[2 * t + 1] = d;
    __syncthreads();
    if (t < cw) {
        float sa = 0.f, sd = 0.f;
        for (int rr = 0; rr < R; ++rr) { sa += scr[2 * (rr * cw + t)]; sd += scr[2 * (rr * cw + t) + 1]; }
        cs[t] = sa;
        cq[t] = sd;
    }
    __syncthreads();
    if (t < gpb) {
        double sa = 0.0, sd = 0.0;
        for (int k = 0; k < cpg; ++k) { sa += cs[t * cpg + k]; sd += cq[t * cpg + k]; }
        // mean and E[x^2] - mean^2 in double (the subtraction cancels); the reciprocal square root in fp32 with one Newton step
        // (~1e-7 relative) instead of a double division and a double square root - software sequences of a few hundred
        // instructions each on the critical path of a launch that lasts 5 us
        const double mean = sa * inv_n;
        double var = sd * inv_n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float vf = (float)var + eps;
        float r = __builtin_amdgcn_rsqf(vf);
        r = r * (1.5f - 0.5f * vf * r * r);
        gm[t] = (float)mean;
        gr[t] = r;
        // fourth part of the table: (mean, rstd) per (image, group), read by the backward pass (hsidm_gn_act_bwd)
        float2* mr = reinterpret_cast<float2*>(reinterpret_cast<float*>(ab) + (size_t)4 * gridDim.y * C);
        mr[(size_t)b * groups + blockIdx.x * gpb + t] = make_float2(gm[t], gr[t]);
    }
    __syncthreads();
    if (t < cw) {
        const int c = c0 + t, g = t / cpg;
        const float sc = gr[g] * gamma[c];
        const float sh = beta[c] - gm[g] * sc;
        ab[(size_t)b * C + c] = make_float2(sc, sh);
        // second half of the table: the same pair as fp16x2, the form the bf16 conv kernels consume (no conversion, and no
        // wait on a parameter load, between requesting a chunk and staging it)
        // The fp16 shift is formed with the ROUNDED scale: scale' * x + (beta - mean * scale') = scale' * (x - mean) + beta, so the
        // scale's rounding error multiplies the deviation from the group mean, not the value itself (a map whose mean is ten
        // standard deviations from zero would otherwise see ten times the error; it matters for the fp16 mode, whose operands
        // are as precise as these pairs).
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const _Float16 sc_h = (_Float16)sc;
        const h2 hh = {sc_h, (_Float16)(beta[c] - gm[g] * (float)sc_h)};
        reinterpret_cast<unsigned*>(ab + (size_t)gridDim.y * C)[(size_t)b * C + c] = __builtin_bit_cast(unsigned, hh);
        // third part: log2(e) * (scale, shift), for the kernels that follow the affine map with SiLU: u = log2(e) * y feeds
        // v_exp_f32 directly (y * sigmoid(y) = u / (1 + 2^-u) / log2(e)); they undo the factor on their fp32 accumulators
        const _Float16 sl_h = (_Float16)(sc * 1.44269504f);
        const h2 hs = {sl_h, (_Float16)(beta[c] * 1.44269504f - gm[g] * (float)sl_h)};
        reinterpret_cast<unsigned*>(ab + (size_t)gridDim.y * C)[(size_t)(gridDim.y + b) * C + c] = __builtin_bit_cast(unsigned, hs);
    }
}

// ---------------------------------------------------------------------------------------------------
// noise_film: grid (ceil(F/256), B).  Every workgroup recomputes the 64 -> 256 -> 64 MLP of its image
// (32 K MACs) in LDS and then emits 256 FiLM outputs.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void noise_film_kernel(const float* __restrict__ gamma, const float* __restrict__ level,
                                                         const int32_t* __restrict__ t_ptr, const float* __restrict__ t_emb,
                                                         float* __restrict__ t_out, int dim,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ wf, const float* __restrict__ bf,
                                                         int F, float* __restrict__ film) {
    extern __shared__ float sm[];          // [dim] pe, [4*dim] hidden, [dim] t
    float* pe = sm;
    float* hid = sm + dim;
    float* te = hid + 4 * dim;
    const int b = blockIdx.y, t = threadIdx.x;
    if (t_emb) {                                  // embedding supplied by the caller (module-level API)
        for (int k = t; k < dim; k += 256) te[k] = t_emb[(size_t)b * dim + k];
        __syncthreads();
    } else {
    const float g = gamma ? gamma[b] : level[*t_ptr + 1];
    const int half = dim >> 1;
    for (int i = t; i < dim; i += 256) {
        const int k = i < half ? i : i - half;
        const float step = (float)k / (float)half;
        const float e = g * expf(-9.210340371976184f * step);      // ln(1e4)
        pe[i] = i < half ? sinf(e) : cosf(e);
    }
    __syncthreads();
    for (int j = t; j < 4 * dim; j += 256) {
        float a = b1[j];
        for (int i = 0; i < dim; ++i) a = fmaf(w1[(size_t)j * dim + i], pe[i], a);
        hid[j] = a / (1.0f + expf(-a));
    }
    __syncthreads();
    for (int k = t; k < dim; k += 256) {
        float a = b2[k];
        for (int j = 0; j < 4 * dim; ++j) a = fmaf(w2[(size_t)k * 4 * dim + j], hid[j], a);
        te[k] = a;
        if (t_out && blockIdx.x == 0) t_out[(size_t)b * dim + k] = a;
    }
    __syncthreads();
    }
    const int f = blockIdx.x * 256 + t;
    if (f < F) {
        float a = bf[f];
        for (int k = 0; k < dim; ++k) a = fmaf(wf[(size_t)f * dim + k], te[k], a);
        film[(size_t)b * F + f] = a;
    }
}

// ---------------------------------------------------------------------------------------------------
// ca_vector: grid (B).  CALayer squeeze-excite vector from per-channel sums.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ca_vector_kernel(const float2* __restrict__ part, int nsplit, int C, int HW, int R,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ ca) {
    extern __shared__ float sm[];          // [C] mean, [R] hidden
    float* mean = sm;
    float* hid = sm + C;
    const int b = blockIdx.x, t = threadIdx.x;
    for (int c = t; c < C; c += 256) {
        float a = 0.f;
        for (int s = 0; s < nsplit; ++s) a += part[((size_t)b * nsplit + s) * C + c].x;
        mean[c] = a / (float)HW;
    }
    __syncthreads();
    for (int r = t; r < R; r += 256) {
        float a = b1[r];
        for (int c = 0; c < C; ++c) a = fmaf(w1[(size_t)r * C + c], mean[c], a);
        hid[r] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float a = b2[c];
        for (int r = 0; r < R; ++r) a = fmaf(w2[(size_t)c * R + r], hid[r], a);
        ca[(size_t)b * C + c] = 1.0f / (1.0f + expf(-a));
    }
}

}  // namespace hsidm

using namespace hsidm;

extern "C" int hsidm_gn_partial(int prec, const void* src0, const void* src1, int C0, int C1, int B, int HW,
                                int nsplit, float* part, void* stream) {
    const int C = C0 + C1;
    if (!src0 || !part || C0 <= 0 || (C0 & 7) || (C1 & 7) || C > 2048 || B <= 0 || HW <= 0 || nsplit <= 0) return HSIDM_E_BADARG;
    if (C1 > 0 && !src1) return HSIDM_E_BADARG;
    dim3 grid(nsplit, B);
    hipStream_t s = (hipStream_t)stream;
    if (prec == HSIDM_BF16)
        hipLaunchKernelGGL(gn_partial_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src0, (const bf16*)src1, C0, C1, HW, nsplit, (float2*)part);
    else if (prec == HSIDM_F16)
        hipLaunchKernelGGL(gn_partial_kernel<f16>, grid, dim3(256), 0, s, (const f16*)src0, (const f16*)src1, C0, C1, HW, nsplit, (float2*)part);
    else if (prec == HSIDM_F32X3)
        hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)src0, (const float*)src1, C0, C1, HW, nsplit, (float2*)part);
    else
        return HSIDM_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int hsidm_gn_finalize(const float* part0, int nsplit0, int C0, const float* part1, int nsplit1, int C1,
                                 int B, int HW, int groups, const float* gamma, const float* beta, float eps,
                                 float* gn_ab, void* stream) {
    const int C = C0 + C1;
    if (!part0 || !gamma || !beta || !gn_ab || groups <= 0 || C0 <= 0 || C1 < 0 || C % groups || nsplit0 <= 0) return HSIDM_E_BADARG;
    if (C1 > 0 && (!part1 || nsplit1 <= 0)) return HSIDM_E_BADARG;
    const int cpg = C / groups;
    if (cpg > 256) return HSIDM_E_UNSUPPORTED;
    // groups per workgroup: as few as possible (more workgroups) while a workgroup's channels fit 256 threads
    int gpb = 1;
    while (groups % (gpb * 2) == 0 && groups / (gpb * 2) >= 8 && gpb * 2 * cpg <= 256) gpb *= 2;
    if (groups / gpb > 8 && gpb * cpg < 8) { while (groups % (gpb * 2) == 0 && gpb * 2 * cpg <= 8) gpb *= 2; }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups / gpb, B), dim3(256), 0, (hipStream_t)stream, (const float2*)part0, nsplit0,
                       C0, (const float2*)part1, nsplit1, C1, HW, groups, gpb, gamma, beta, eps, 1.0 / ((double)cpg * HW), (float2*)gn_ab);
    return (int)hipGetLastError();
}

extern "C" int hsidm_noise_film(const float* gamma, const float* level_table, const int32_t* t_ptr, const float* t_emb,
                                int B, int dim, const float* w1, const float* b1, const float* w2, const float* b2,
                                const float* wf, const float* bf, int F, float* film, float* t_out, void* stream) {
    if (!t_emb && ((!gamma && !(level_table && t_ptr)) || !w1 || !b1 || !w2 || !b2)) return HSIDM_E_BADARG;
    if (!wf || !bf || !film) return HSIDM_E_BADARG;
    if (dim <= 0 || (dim & 1) || dim > 1024 || F <= 0 || B <= 0) return HSIDM_E_BADARG;
    dim3 grid((F + 255) / 256, B);
    const size_t lds = (size_t)(6 * dim) * sizeof(float);
    hipLaunchKernelGGL(noise_film_kernel, grid, dim3(256), lds, (hipStream_t)stream, gamma, level_table, t_ptr, t_emb, t_out, dim,
                       w1, b1, w2, b2, wf, bf, F, film);
    return (int)hipGetLastError();
}

extern "C" int hsidm_ca_vector(const float* part, int nsplit, int B, int C, int HW, int R,
                               const float* w1, const float* b1, const float* w2, const float* b2,
                               float* ca, void* stream) {
    if (!part || !w1 || !b1 || !w2 || !b2 || !ca || C <= 0 || R <= 0 || C > 4096) return HSIDM_E_BADARG;
    const size_t lds = (size_t)(C + R) * sizeof(float);
    hipLaunchKernelGGL(ca_vector_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, (const float2*)part, nsplit, C, HW, R,
                       w1, b1, w2, b2, ca);
    return (int)hipGetLastError();
}
