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

// element type codes of the three arrays: 0 bf16, 1 fp32, 2 fp16
__device__ __forceinline__ float ld_elem(const void* p, int64_t i, int is_f32) {
    return is_f32 == 1 ? reinterpret_cast<const float*>(p)[i]
                       : (is_f32 == 2 ? (float)reinterpret_cast<const f16*>(p)[i] : (float)reinterpret_cast<const bf16*>(p)[i]);
}

// C[b] (M x N, row-major, unit column stride) = alpha * A[b] (M x K) * B[b] (K x N).
// Workgroup = one 32 x 32 tile of C; its four waves split K by 32-deep chunks (wave w takes chunks w, w + 4, ...) and meet in LDS
// at the end.  Every wave is a stream of its own - private LDS staging tiles, no workgroup barrier inside the K loop, the
// operands of its next chunk requested before the current one is multiplied: the attention backward's products are 256 x 256
// ... 512 per image at batch 4 with K = 256 or 512, i.e. two to four chunks per wave, two of them in flight at any time.  (The first version
// shared every chunk between the four waves: two barriers and one exposed L2 / HBM round trip per chunk, 32 us per launch for
// 0.27 GFLOP; the products are latency, not arithmetic.)
__global__ __launch_bounds__(256, 2) void bgemm_kernel(GemmParams p) {
    constexpr int KC = 32, PITCH = 33, NL = KC * 32 / 64;        // NL elements per lane, operand and chunk
    extern __shared__ float bg_lds[];                           // per wave: As[KC][PITCH], Bs[KC][PITCH]; afterwards red[3][32][33]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* As = bg_lds + wave * (2 * KC * PITCH);
    float* Bs = As + KC * PITCH;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, bz = blockIdx.z;
    const int64_t abase = (int64_t)bz * p.sab, bbase = (int64_t)bz * p.sbb;
    // consecutive lanes walk the unit-stride axis of each operand
    const bool a_kfast = p.sak == 1, b_kfast = p.sbk == 1;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const int nchunks = (p.K + KC - 1) / KC;
    float ra[2][NL], rb[2][NL];
    // element i of a lane's share of a chunk: (fast, slow) index = (lane % 32, lane / 32 + 2 i), fast = k when the operand's unit
    // stride is k, else the row / column - either way the address is linear in i and the bounds test is i < a per-lane limit
    const int lf = lane & 31, ls = lane >> 5;
    auto coords = [&](bool kfast, int i, int& k, int& r) __attribute__((always_inline)) {
        if (kfast) { k = lf; r = ls + 2 * i; } else { r = lf; k = ls + 2 * i; }
    };
    // every load is unconditional (out-of-range elements re-read the lane's last valid one, or the batch's first element, and
    // are zeroed afterwards): a guarded load is a branch and a wait of its own, and 32 of those in sequence were the kernel
    auto fetch = [&](float (&dst)[NL], const void* base, int is_f32, int64_t safe, int64_t off0, int64_t step, int lim) __attribute__((always_inline)) {
        const int64_t o = lim > 0 ? off0 : safe;
        const int last = lim > 0 ? lim - 1 : 0;
        if (is_f32 == 1) {
            const float* q = reinterpret_cast<const float*>(base) + o;
#pragma unroll
            for (int i = 0; i < NL; ++i) dst[i] = q[(i < last ? i : last) * step];
        } else if (is_f32 == 2) {
            const f16* q = reinterpret_cast<const f16*>(base) + o;
#pragma unroll
            for (int i = 0; i < NL; ++i) dst[i] = (float)q[(i < last ? i : last) * step];
        } else {
            const bf16* q = reinterpret_cast<const bf16*>(base) + o;
#pragma unroll
            for (int i = 0; i < NL; ++i) dst[i] = (float)q[(i < last ? i : last) * step];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) dst[i] = i < lim ? dst[i] : 0.f;
    };
    auto issue = [&](int S, int chunk) __attribute__((always_inline)) {
        const int k0 = chunk * KC;
        {
            const int kl = a_kfast ? lf : ls, rl = a_kfast ? ls : lf;
            const int lim = (k0 + kl < p.K && m0 + rl < p.M) ? ((a_kfast ? p.M - m0 - rl : p.K - k0 - kl) + 1) / 2 : 0;
            fetch(ra[S], p.a, p.a_f32, abase, abase + (int64_t)(m0 + rl) * p.sam + (int64_t)(k0 + kl) * p.sak, 2 * (a_kfast ? p.sam : p.sak), lim);
        }
        {
            const int kl = b_kfast ? lf : ls, rl = b_kfast ? ls : lf;
            const int lim = (k0 + kl < p.K && n0 + rl < p.N) ? ((b_kfast ? p.N - n0 - rl : p.K - k0 - kl) + 1) / 2 : 0;
            fetch(rb[S], p.b, p.b_f32, bbase, bbase + (int64_t)(k0 + kl) * p.sbk + (int64_t)(n0 + rl) * p.sbn, 2 * (b_kfast ? p.sbn : p.sbk), lim);
        }
    };
    auto commit = [&](int S) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int k, m;
            coords(a_kfast, i, k, m);
            As[k * PITCH + m] = ra[S][i];
            int kb, n;
            coords(b_kfast, i, kb, n);
            Bs[kb * PITCH + n] = rb[S][i];
        }
    };
    auto multiply = [&]() __attribute__((always_inline)) {
        const int h = lane >> 5, r = lane & 31;
#pragma unroll 8
        for (int kk = 0; kk < KC; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(kk + h) * PITCH + r], Bs[(kk + h) * PITCH + r], acc, 0, 0, 0);
    };
    // the wave's LDS accesses execute in order; the fences keep the compiler from moving them across the hand-over points
    auto wave_sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    int c = wave;
    if (c < nchunks) issue(0, c);
    if (c + 4 < nchunks) issue(1, c + 4);
    for (int s = 0; c < nchunks; c += 4, s ^= 1) {
        wave_sync();                                            // the previous chunk's fragment reads are done
        if (s == 0) commit(0); else commit(1);
        wave_sync();
        if (c + 8 < nchunks) { if (s == 0) issue(0, c + 8); else issue(1, c + 8); }
        multiply();
    }
    // the four K streams meet in a fixed order (wave 0 + 1 + 2 + 3): deterministic
    __syncthreads();                                            // every wave is done with its staging tiles: red aliases them
    float (*red)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(bg_lds);
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
                if (p.c_f32 == 1) reinterpret_cast<float*>(p.c)[idx] = v;
                else if (p.c_f32 == 2) reinterpret_cast<f16*>(p.c)[idx] = (f16)Elem<f16>::sat(v);
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
    constexpr size_t lds = (size_t)4 * 2 * 32 * 33 * sizeof(float);
    static PerDeviceOnce once;
    if (int rc = raise_lds_cap(once, &bgemm_kernel, lds)) return rc;
    hipLaunchKernelGGL(bgemm_kernel, dim3((N + 31) / 32, (M + 31) / 32, batch), dim3(256), lds, (hipStream_t)stream, p);
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
