// Single-head self-attention core on the matrix cores: O = softmax(Q K^T / sqrt(C)) V.
//
// Replaces the two einsums + softmax of SelfAttention.forward (reference unet.py:130-140).  The
// sequence is a feature map of at most 32x32 pixels (N <= 1024; 256 at the shipped config), so one
// workgroup keeps a whole 32-query score panel S[32][N] in LDS and does an exact (not online) row
// softmax in fp32 - the same arithmetic as the reference.
//
//   grid = (ceil(N/32), B), 256 threads = 4 waves.
//   phase 1  S = Q K^T : Q panel in LDS (bf16 hi[,lo]); wave w owns key tiles w, w+4, ...; K rows are
//            read straight from global as MFMA B-fragments (each K element is used once per workgroup).
//   phase 2  row softmax: wave w owns rows 8w..8w+7, wave-shuffle max / sum.
//   phase 3  O = P V : wave w owns channel tiles w, w+4, ...; each 32x32 V tile is transposed through
//            a wave-private LDS patch so the key axis becomes the contiguous MFMA k axis.
// qkv is NHWC [B][N][3C]; q, k, v are the channel thirds (reference unet.py:129, n_head = 1).
#include "common.h"
#include "../../include/hsidm.h"

namespace hsidm {

template <bool SPLIT> struct Frag { bf16x8 hi; bf16x8 lo; };

template <bool SPLIT>
__device__ __forceinline__ void make_frag(const float (&v)[8], Frag<SPLIT>& f) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f.hi[k] = (bf16)v[k];
        if (SPLIT) f.lo[k] = (bf16)(v[k] - (float)f.hi[k]);
    }
}

template <bool SPLIT>
__device__ __forceinline__ void mma(f32x16& acc, const Frag<SPLIT>& a, const Frag<SPLIT>& b) {
    if (SPLIT) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

template <typename ActT, bool SPLIT>
__global__ __launch_bounds__(256) void attention_kernel(const ActT* __restrict__ qkv, ActT* __restrict__ out,
                                                        int N, int C, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int QS = C + 8;                       // bf16 elements per Q row (+16 B pad)
    const int NP = (N + 31) & ~31;              // keys padded to the MFMA tile
    const int SS = NP + 4;                      // fp32 elements per score row
    constexpr int VS = 40;                      // bf16 elements per transposed-V row (32 keys + pad)
    // LDS carve: [Q hi | Q lo] then S; the Q region is reused for the per-wave V^T patches.
    bf16* q_hi = reinterpret_cast<bf16*>(smem_raw);
    bf16* q_lo = q_hi + (SPLIT ? 32 * QS : 0);
    const size_t q_bytes = (size_t)(SPLIT ? 2 : 1) * 32 * QS * 2;
    const size_t vt_bytes = (size_t)4 * (SPLIT ? 2 : 1) * 32 * VS * 2;
    const size_t reg0 = ((q_bytes > vt_bytes ? q_bytes : vt_bytes) + 15) & ~(size_t)15;
    float* S = reinterpret_cast<float*>(smem_raw + reg0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 32;
    const size_t row3 = (size_t)3 * C;
    const ActT* base = qkv + (size_t)b * N * row3;

    // ---- stage the Q panel -----------------------------------------------------------------------
    const int nvec = C >> 3;
    for (int i = tid; i < 32 * nvec; i += 256) {
        const int r = i / nvec, cvi = i - r * nvec;
        float v[8];
        if (q0 + r < N) Vec8<ActT>::load(base + (size_t)(q0 + r) * row3 + cvi * 8, v);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        Frag<SPLIT> f;
        make_frag<SPLIT>(v, f);
        *reinterpret_cast<bf16x8*>(q_hi + r * QS + cvi * 8) = f.hi;
        if (SPLIT) *reinterpret_cast<bf16x8*>(q_lo + r * QS + cvi * 8) = f.lo;
    }
    __syncthreads();

    // ---- S = scale * Q K^T -----------------------------------------------------------------------------
    const int nkt = NP >> 5;
    for (int kt = wave; kt < nkt; kt += 4) {
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const int key = kt * 32 + lr;
        const bool kok = key < N;
        const ActT* krow = base + (size_t)(kok ? key : 0) * row3 + C + 8 * lh;
        for (int c16 = 0; c16 < C; c16 += 16) {
            Frag<SPLIT> a, bb;
            a.hi = *reinterpret_cast<const bf16x8*>(q_hi + lr * QS + c16 + 8 * lh);
            if (SPLIT) a.lo = *reinterpret_cast<const bf16x8*>(q_lo + lr * QS + c16 + 8 * lh);
            float v[8];
            Vec8<ActT>::load(krow + c16, v);
            if (!kok) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
            make_frag<SPLIT>(v, bb);
            mma<SPLIT>(acc, a, bb);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
            S[row * SS + key] = kok ? acc[j] * scale : -INFINITY;
        }
    }
    __syncthreads();

    // ---- exact row softmax (fp32), P overwrites S ------------------------------------------------------
    for (int r = wave * 8; r < wave * 8 + 8; ++r) {
        float* srow = S + r * SS;
        float m = -INFINITY;
        for (int k = lane; k < NP; k += 64) m = fmaxf(m, srow[k]);
        m = wave_max(m);
        float sum = 0.f;
        for (int k = lane; k < NP; k += 64) {
            const float e = __expf(srow[k] - m);
            srow[k] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int k = lane; k < NP; k += 64) srow[k] *= inv;
    }
    __syncthreads();

    // ---- O = P V ------------------------------------------------------------------------------------------
    bf16* vt_hi = reinterpret_cast<bf16*>(smem_raw) + (size_t)wave * (SPLIT ? 2 : 1) * 32 * VS;
    bf16* vt_lo = vt_hi + (SPLIT ? 32 * VS : 0);
    const int nct = C >> 5;
    const int iters = (nct + 3) >> 2;
    for (int it = 0; it < iters; ++it) {
        const int ct = it * 4 + wave;
        const bool active = ct < nct;
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();                                   // previous patch fully consumed
            if (active) {
                // 32 keys x 32 channels = 128 vectors of 8 channels; 2 per lane
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int vi = lane + 64 * h;
                    const int kr = vi >> 2, cq = vi & 3;        // key row, channel quad (8 channels)
                    const int key = kt * 32 + kr;
                    float v[8];
                    if (key < N) Vec8<ActT>::load(base + (size_t)key * row3 + 2 * C + ct * 32 + cq * 8, v);
                    else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bf16 hi = (bf16)v[k];
                        vt_hi[(cq * 8 + k) * VS + kr] = hi;
                        if (SPLIT) vt_lo[(cq * 8 + k) * VS + kr] = (bf16)(v[k] - (float)hi);
                    }
                }
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    float pv[8];
                    const float* prow = S + lr * SS + kt * 32 + kk * 16 + 8 * lh;
#pragma unroll
                    for (int k = 0; k < 8; ++k) pv[k] = prow[k];
                    Frag<SPLIT> a, bb;
                    make_frag<SPLIT>(pv, a);
                    bb.hi = *reinterpret_cast<const bf16x8*>(vt_hi + lr * VS + kk * 16 + 8 * lh);
                    if (SPLIT) bb.lo = *reinterpret_cast<const bf16x8*>(vt_lo + lr * VS + kk * 16 + 8 * lh);
                    mma<SPLIT>(acc, a, bb);
                }
            }
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
                if (q0 + row < N) out[((size_t)b * N + q0 + row) * C + ct * 32 + lr] = from_f32<ActT>(acc[j]);
            }
        }
    }
}

template <typename ActT, bool SPLIT>
static int launch_attention(const void* qkv, void* out, int B, int N, int C, hipStream_t s) {
    const int QS = C + 8, NP = (N + 31) & ~31, SS = NP + 4;
    const size_t q_bytes = (size_t)(SPLIT ? 2 : 1) * 32 * QS * 2;
    const size_t vt_bytes = (size_t)4 * (SPLIT ? 2 : 1) * 32 * 40 * 2;
    const size_t reg0 = ((q_bytes > vt_bytes ? q_bytes : vt_bytes) + 15) & ~(size_t)15;
    const size_t lds = reg0 + (size_t)32 * SS * 4;
    if (lds > 160 * 1024) return HSIDM_E_UNSUPPORTED;
    static bool attr_done = false;              // raise the dynamic-LDS cap once (not a stream operation)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<ActT, SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    dim3 grid((N + 31) / 32, B);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(attention_kernel<ActT, SPLIT>), grid, dim3(256), lds, s, (const ActT*)qkv, (ActT*)out, N, C,
                       1.0f / sqrtf((float)C));
    return (int)hipGetLastError();
}

}  // namespace hsidm

extern "C" int hsidm_attention(int prec, const void* qkv, void* out, int B, int N, int C, void* stream) {
    if (!qkv || !out || B <= 0 || N <= 0 || C <= 0 || (C & 31)) return HSIDM_E_BADARG;
    if (N > 1024) return HSIDM_E_UNSUPPORTED;
    if (prec == HSIDM_BF16) return hsidm::launch_attention<hsidm::bf16, false>(qkv, out, B, N, C, (hipStream_t)stream);
    if (prec == HSIDM_F32X3) return hsidm::launch_attention<float, true>(qkv, out, B, N, C, (hipStream_t)stream);
    return HSIDM_E_BADARG;
}
