// Strided batched GEMM on the exact fp32 matrix instruction + row softmax forward / backward: the pieces of the self-attention
// backward pass (reference unet.py:130-140 differentiated by autograd in model/model.py:55):
//   S = scale * Q K^T, P = softmax(S);  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - rowsum(dP o P)) * scale;  dQ = dS K;  dK = dS^T Q
// q, k, v (and dq, dk, dv) are channel thirds of the NHWC qkv tensor, so every operand is addressed by element strides and no
// slice or transpose is ever materialised.  N <= 1024 tokens and C = 512: under 2 % of a training step's arithmetic, so one
// kernel serves both precision modes: operands are widened to fp32 on their way into LDS and multiplied by
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation: bit-for-bit an ordered fmaf chain).
#include "common.h"
#include "../../include/hsidm.h"

namespace hsidm {

struct GemmParams {
    const void* a; const void* b; void* c;
    int64_t sab, sam, sak, sbb, sbk, sbn, scb, scm;
    int M, N, K;
    float alpha;
    int a_f32, b_f32, c_f32;
};

__device__ __forceinline__ float ld_elem(const void* p, int64_t i, int is_f32) {
    return is_f32 ? reinterpret_cast<const float*>(p)[i] : (float)reinterpret_cast<const bf16*>(p)[i];
}

// C[b] (M x N, row-major, unit column stride) = alpha * A[b] (M x K) * B[b] (K x N).
// Workgroup = one 32 x 32 tile of C, its four waves split every 64-deep K chunk four ways (16 k each = 8 MFMAs) and meet in
// LDS at the end: the attention backward's products are 256 x 256 ... 512 per image at batch 4, so 64 x 64 tiles would be 64
// workgroups on 256 CUs; 32 x 32 tiles with the K split inside the workgroup give 256-512 of them at the same staging cost.
__global__ __launch_bounds__(256) void bgemm_kernel(GemmParams p) {
    constexpr int KC = 64, PITCH = 33;
    __shared__ float As[KC * PITCH], Bs[KC * PITCH];
    __shared__ float red[3][32 * 33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, bz = blockIdx.z;
    const int64_t abase = (int64_t)bz * p.sab, bbase = (int64_t)bz * p.sbb;
    // consecutive threads walk the unit-stride axis of each operand
    const bool a_kfast = p.sak == 1, b_kfast = p.sbk == 1;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    float ra[8], rb[8];
    auto issue = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int k, m;
            if (a_kfast) { k = tid & 63; m = (tid >> 6) + 4 * i; } else { m = tid & 31; k = (tid >> 5) + 8 * i; }
            ra[i] = (m0 + m < p.M && k0 + k < p.K) ? ld_elem(p.a, abase + (int64_t)(m0 + m) * p.sam + (int64_t)(k0 + k) * p.sak, p.a_f32) : 0.f;
            int kb, n;
            if (b_kfast) { kb = tid & 63; n = (tid >> 6) + 4 * i; } else { n = tid & 31; kb = (tid >> 5) + 8 * i; }
            rb[i] = (n0 + n < p.N && k0 + kb < p.K) ? ld_elem(p.b, bbase + (int64_t)(k0 + kb) * p.sbk + (int64_t)(n0 + n) * p.sbn, p.b_f32) : 0.f;
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int k, m;
            if (a_kfast) { k = tid & 63; m = (tid >> 6) + 4 * i; } else { m = tid & 31; k = (tid >> 5) + 8 * i; }
            As[k * PITCH + m] = ra[i];
            int kb, n;
            if (b_kfast) { kb = tid & 63; n = (tid >> 6) + 4 * i; } else { n = tid & 31; kb = (tid >> 5) + 8 * i; }
            Bs[kb * PITCH + n] = rb[i];
        }
    };
    issue(0);
    for (int k0 = 0; k0 < p.K; k0 += KC) {
        __syncthreads();
        commit();
        __syncthreads();
        if (k0 + KC < p.K) issue(k0 + KC);
        const int h = lane >> 5, r = lane & 31;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const int k = 16 * wave + kk + h;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * PITCH + r], Bs[k * PITCH + r], acc, 0, 0, 0);
        }
    }
    // the four K quarters meet in a fixed order (wave 0 + 1 + 2 + 3): deterministic
    const int col = lane & 31;
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[wave - 1][((j & 3) + 8 * (j >> 2) + 4 * (lane >> 5)) * 33 + col] = acc[j];
    }
    __syncthreads();
    if (wave == 0) {
        const int n = n0 + col;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
            const int m = m0 + row;
            const float v = p.alpha * (((acc[j] + red[0][row * 33 + col]) + red[1][row * 33 + col]) + red[2][row * 33 + col]);
            if (m < p.M && n < p.N) {
                const int64_t idx = (int64_t)bz * p.scb + (int64_t)m * p.scm + n;
                if (p.c_f32) reinterpret_cast<float*>(p.c)[idx] = v;
                else reinterpret_cast<bf16*>(p.c)[idx] = (bf16)v;
            }
        }
    }
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// one workgroup per row, N <= 1024: s <- softmax(s)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int N) {
    __shared__ float red[4];
    float* row = s + (size_t)blockIdx.x * N;
    float v[4];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = threadIdx.x + 256 * i;
        v[i] = j < N ? row[j] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    m = block_reduce(m, red, true);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (threadIdx.x + 256 * i) < N ? expf(v[i] - m) : 0.f; sum += v[i]; }
    sum = block_reduce(sum, red, false);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = threadIdx.x + 256 * i;
        if (j < N) row[j] = v[i] * inv;
    }
}
// dp <- p o (dp - rowsum(dp o p)) * scale
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ p, float* __restrict__ dp, int N, float scale) {
    __shared__ float red[4];
    const float* pr = p + (size_t)blockIdx.x * N;
    float* dr = dp + (size_t)blockIdx.x * N;
    float pv[4], dv[4];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = threadIdx.x + 256 * i;
        pv[i] = j < N ? pr[j] : 0.f;
        dv[i] = j < N ? dr[j] : 0.f;
        dot = fmaf(pv[i], dv[i], dot);
    }
    dot = block_reduce(dot, red, false);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = threadIdx.x + 256 * i;
        if (j < N) dr[j] = pv[i] * (dv[i] - dot) * scale;
    }
}

}  // namespace hsidm

using namespace hsidm;

extern "C" int hsidm_bgemm(const void* a, int a_f32, int64_t sab, int64_t sam, int64_t sak, const void* b, int b_f32, int64_t sbb,
                           int64_t sbk, int64_t sbn, void* c, int c_f32, int64_t scb, int64_t scm, int M, int N, int K, int batch,
                           float alpha, void* stream) {
    if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535) return HSIDM_E_BADARG;
    GemmParams p;
    p.a = a; p.b = b; p.c = c;
    p.sab = sab; p.sam = sam; p.sak = sak; p.sbb = sbb; p.sbk = sbk; p.sbn = sbn; p.scb = scb; p.scm = scm;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.a_f32 = a_f32; p.b_f32 = b_f32; p.c_f32 = c_f32;
    hipLaunchKernelGGL(bgemm_kernel, dim3((N + 31) / 32, (M + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int hsidm_softmax_rows(float* s, int64_t rows, int N, void* stream) {
    if (!s || rows <= 0 || N <= 0 || N > 1024) return HSIDM_E_BADARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, N);
    return (int)hipGetLastError();
}

extern "C" int hsidm_softmax_bwd_rows(const float* p, float* dp, int64_t rows, int N, float scale, void* stream) {
    if (!p || !dp || rows <= 0 || N <= 0 || N > 1024) return HSIDM_E_BADARG;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, N, scale);
    return (int)hipGetLastError();
}
