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
#include <cstdlib>
#include "../../include/hsidm.h"

namespace hsidm {

// Op: the 16-bit MFMA operand type (bf16 | fp16); SPLIT (fp32 storage): operands as Op hi + lo, three MFMAs per product
template <typename Op, bool SPLIT> struct Frag { typename Elem<Op>::x8 hi; typename Elem<Op>::x8 lo; };

template <typename Op, bool SPLIT>
__device__ __forceinline__ void make_frag(const float (&v)[8], Frag<Op, SPLIT>& f) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f.hi[k] = (Op)v[k];
        if (SPLIT) f.lo[k] = (Op)(v[k] - (float)f.hi[k]);
    }
}

template <typename Op, bool SPLIT>
__device__ __forceinline__ void mma(f32x16& acc, const Frag<Op, SPLIT>& a, const Frag<Op, SPLIT>& b) {
    if (SPLIT) {
        acc = Elem<Op>::mfma(a.lo, b.hi, acc);
        acc = Elem<Op>::mfma(a.hi, b.lo, acc);
    }
    acc = Elem<Op>::mfma(a.hi, b.hi, acc);
}

template <typename ActT, bool SPLIT, typename Op = bf16>
__global__ __launch_bounds__(256) void attention_kernel(const ActT* __restrict__ qkv, ActT* __restrict__ out,
                                                        int N, int C, float scale) {
    using bf16 = Op;                                 // (the body below names its operand type bf16)
    using bf16x8 = typename Elem<Op>::x8;
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
        Frag<Op, SPLIT> f;
        make_frag<Op, SPLIT>(v, f);
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
            Frag<Op, SPLIT> a, bb;
            a.hi = *reinterpret_cast<const bf16x8*>(q_hi + lr * QS + c16 + 8 * lh);
            if (SPLIT) a.lo = *reinterpret_cast<const bf16x8*>(q_lo + lr * QS + c16 + 8 * lh);
            float v[8];
            Vec8<ActT>::load(krow + c16, v);
            if (!kok) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
            make_frag<Op, SPLIT>(v, bb);
            mma<Op, SPLIT>(acc, a, bb);
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
                    Frag<Op, SPLIT> a, bb;
                    make_frag<Op, SPLIT>(pv, a);
                    bb.hi = *reinterpret_cast<const bf16x8*>(vt_hi + lr * VS + kk * 16 + 8 * lh);
                    if (SPLIT) bb.lo = *reinterpret_cast<const bf16x8*>(vt_lo + lr * VS + kk * 16 + 8 * lh);
                    mma<Op, SPLIT>(acc, a, bb);
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

template <typename ActT, bool SPLIT, typename Op = bf16>
static int launch_attention(const void* qkv, void* out, int B, int N, int C, hipStream_t s) {
    const int QS = C + 8, NP = (N + 31) & ~31, SS = NP + 4;
    const size_t q_bytes = (size_t)(SPLIT ? 2 : 1) * 32 * QS * 2;
    const size_t vt_bytes = (size_t)4 * (SPLIT ? 2 : 1) * 32 * 40 * 2;
    const size_t reg0 = ((q_bytes > vt_bytes ? q_bytes : vt_bytes) + 15) & ~(size_t)15;
    const size_t lds = reg0 + (size_t)32 * SS * 4;
    if (lds > 160 * 1024) return HSIDM_E_UNSUPPORTED;
    static PerDeviceOnce once;                  // raise the dynamic-LDS cap once per device (not a stream operation)
    if (int rc = raise_lds_cap(once, &attention_kernel<ActT, SPLIT, Op>, 160 * 1024)) return rc;
    dim3 grid((N + 31) / 32, B);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(attention_kernel<ActT, SPLIT, Op>), grid, dim3(256), lds, s, (const ActT*)qkv, (ActT*)out, N, C,
                       1.0f / sqrtf((float)C));
    return (int)hipGetLastError();
}


// ---- attention_v2 (bf16 mode, N = 32*NKT keys, C % 64 == 0): the whole row of scores lives in registers ----------------
// Measured on the kernel above (batch 120, N = 256, C = 512): 155 us per launch = 104 TFLOP/s; K rows loaded as MFMA
// fragments straight from global (32 rows per wave-load, texture-addresser bound), V transposed with 2-byte LDS writes
// behind two barriers per 32 keys, and every 32-query workgroup re-reading all of K and V (8x per image).  Here
//   * a wave owns 32 queries and ALL keys: S^T = K Q^T (keys on the accumulator rows, the query on the lane), so the row
//     softmax is a per-lane reduction plus one exchange between the two lane halves, exact and in fp32 as before;
//   * the probabilities never leave the registers: an accumulator tile converted to bf16 is directly the B operand of
//     O^T = V^T P^T (cdna_hip_programming.md, "an accumulator tile as the next MFMA's operand"); its k order is permuted
//     (element j of lane half h = key 16s + 8(j>>2) + 4h + (j&3)), and V^T fragments in exactly that order come from a
//     row-major V image in LDS through ds_read_b64_tr_b16 (4 keys x 16 channels per 16-lane group, delivered
//     channel-major): two reads per fragment, 192-byte key rows -> conflict-free;
//   * K and V stream through a double-buffered LDS image in 64-channel chunks with row-contiguous 16-byte loads, shared
//     by the 4 waves (128 queries) of the workgroup: K and V are read twice per image instead of eight times;
//   * O^T tiles have the query on the lane and 4 consecutive channels per register quad: 8-byte stores, no epilogue pass.
// Round 6: what bounds it at the benchmark's batch is HBM, not LDS or the matrix pipe - 32 GFLOP over 378 MB (NW = 4: K and V are read by
// both workgroups of an image) or 252 MB (NW = 8: one workgroup per image) is 85 / 128 FLOP per byte against a ridge of 312.  A form in
// which every LDS operand fed two MFMAs (64 queries per wave pair, the probabilities exchanged through LDS; git history: "attention_v3")
// halved the LDS reads and ran 7 % SLOWER; reading K and V once is worth 8 % (profiles/r06_attention/ab_forms.txt).
#ifndef HSIDM_ATT_VRS
#define HSIDM_ATT_VRS 80
#endif
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int V> struct AttTag { static constexpr int value = V; };

// PF = 2 (the 8-wave form; C % 128 == 0): the K / V chunk stream is requested TWO chunks ahead into two register sets (the V chunks
// across the softmax: V chunk 0 and 1 are on their way during the last K iterations) - the launch is bound by what its loads take to
// come back (one workgroup per CU in lockstep bursts), not by the matrix pipe.
template <int NKT, int NW, typename E, int PF = 1>
__global__ __launch_bounds__(64 * NW, (HSIDM_ATT_VRS == 80 && NW < 8) ? 2 : 1) void attention_v2_kernel(const E* __restrict__ qkv, E* __restrict__ out, int C, float scale) {
    using bf16 = E;                                  // (the body below names its element type bf16; E = bf16 | fp16)
    using bf16x8 = typename Elem<E>::x8;
    using bf16x4 = typename Elem<E>::x4;
    constexpr int N = 32 * NKT, T = 64 * NW, NV = N * 8 / T;
    // bf16 per key row: 144 B (ds_read_b128 fragments) / 160 B (transposed reads: the four key rows a 16-lane group touches start
    // 40 banks apart - banks 0, 40, 16, 56, eight each: conflict-free like the 192-byte pitch, and two 256-key buffers are exactly
    // 80 KiB, so that two workgroups fit a CU: 480 workgroups of a 240-image batch are resident at once instead of in two rounds)
    constexpr int KRS = 72, VRS = HSIDM_ATT_VRS;
    constexpr int BUFE = N * VRS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* buf = reinterpret_cast<bf16*>(smem_raw);         // [2][BUFE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 32 * NW + 32 * wave;
    const size_t row3 = (size_t)3 * C;
    const bf16* base = qkv + (size_t)b * N * row3;
    const int nch = C >> 6;

    u32x4 hregs[PF][NV];
    u32x4 (&hreg)[NV] = hregs[0];
    // stream chunk s = 0 .. 2 nch - 1: K chunk s, then V chunk s - nch; register set `set`
    auto issue_s = [&](auto set_tag, int sidx) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value;
        const int third = sidx < nch ? 1 : 2, chunk = sidx < nch ? sidx : sidx - nch;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * T;
            hregs[SET][i] = *reinterpret_cast<const u32x4*>(base + (size_t)(v >> 3) * row3 + third * C + chunk * 64 + (v & 7) * 8);
        }
    };
    auto commit_s = [&](auto set_tag, int bi, int rs) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * T;
            *reinterpret_cast<u32x4*>(buf + bi * BUFE + (v >> 3) * rs + (v & 7) * 8) = hregs[SET][i];
        }
    };
    auto issue = [&](int chunk, int third) __attribute__((always_inline)) { issue_s(AttTag<0>{}, third == 1 ? chunk : nch + chunk); };
    auto commit = [&](int bi, int rs) __attribute__((always_inline)) { commit_s(AttTag<0>{}, bi, rs); };

    // ---- phase 1: S^T[key][query] = K Q^T, all NKT key tiles of this wave's 32 queries ----------------------------------
    const bf16* qrow = base + (size_t)(q0 + lr) * row3 + 8 * lh;
    bf16x8 qf[4], qn[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + kk * 16);
    issue(0, 1);
    if constexpr (PF == 2) issue_s(AttTag<PF - 1>{}, 1);
    commit(0, KRS);
    __syncthreads();
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int j = 0; j < 16; ++j) sc[kt][j] = 0.f;
    auto k_body = [&](int chunk, auto par_tag) __attribute__((always_inline)) {        // PF = 2: stream chunk `chunk` lives in set / buffer PAR
        constexpr int PAR = decltype(par_tag)::value;
        const bool more = chunk + 1 < nch;
        if constexpr (PF == 2) issue_s(AttTag<PAR>{}, chunk + 2);                     // (set PAR held this chunk: committed an iteration ago)
        if (more) {
            if constexpr (PF == 1) issue(chunk + 1, 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) qn[kk] = *reinterpret_cast<const bf16x8*>(qrow + (chunk + 1) * 64 + kk * 16);
        }
        const bf16* kb = buf + (chunk & 1) * BUFE + lr * KRS + 8 * lh;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kb + kt * 32 * KRS + kk * 16);
                sc[kt] = Elem<E>::mfma(a, qf[kk], sc[kt]);
            }
        if constexpr (PF == 2) {
            // the next stream chunk (K chunk + 1, or V chunk 0 behind the last K chunk: V's row pitch) into the other buffer
            commit_s(AttTag<PAR ^ (PF - 1)>{}, (chunk + 1) & 1, more ? KRS : VRS);
        } else if (more) commit((chunk + 1) & 1, KRS);
        if (more) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) qf[kk] = qn[kk];
        }
        __syncthreads();
    };
    if constexpr (PF == 2) {
        for (int chunk = 0; chunk < nch; chunk += 2) { k_body(chunk, AttTag<0>{}); k_body(chunk + 1, AttTag<PF - 1>{}); }
    } else {
        for (int chunk = 0; chunk < nch; ++chunk) k_body(chunk, AttTag<0>{});
    }

    // ---- exact softmax over the keys of this lane's query (fp32); V chunk 0 is on its way meanwhile ---------------------
    if constexpr (PF == 1) issue(0, 2);
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int j = 0; j < 16; ++j) { sc[kt][j] *= scale; m = fmaxf(m, sc[kt][j]); }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int j = 0; j < 16; ++j) { sc[kt][j] = __expf(sc[kt][j] - m); sum += sc[kt][j]; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    bf16x8 pf[NKT][2];                                     // P^T fragments: k-step st of key tile kt = registers 8st .. 8st+7
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[kt][st][j] = (bf16)(sc[kt][8 * st + j] * inv);
    if constexpr (PF == 1) {
        commit(0, VRS);
        __syncthreads();
    }                       // (PF = 2: V chunk 0 went into its buffer behind the last K iteration, before that iteration's barrier)

    // ---- phase 2: O^T[channel][query] = V^T P^T, 64 channels per chunk ------------------------------------------------------
    const int gi = lane & 15;
    const int tr_lane = (4 * lh + (gi >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (gi & 3);   // this lane's address duty in its 16-lane group
    bf16* orow = out + ((size_t)b * N + q0 + lr) * C + 4 * lh;
    auto v_body = [&](int chunk, auto par_tag) __attribute__((always_inline)) {        // (nch is even in the PF = 2 form: V chunk c in set / buffer c & 1)
        constexpr int PAR = decltype(par_tag)::value;
        const bool more = chunk + 1 < nch;
        if constexpr (PF == 2) { if (chunk + 2 < nch) issue_s(AttTag<PAR>{}, nch + chunk + 2); }
        else if (more) issue(chunk + 1, 2);
        const bf16* vb = buf + (chunk & 1) * BUFE + tr_lane;
        f32x16 o[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int j = 0; j < 16; ++j) o[c2][j] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
                    const bf16* a0 = vb + (kt * 32 + st * 16) * VRS + c2 * 32;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0 + 8 * VRS));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[c2] = Elem<E>::mfma(__builtin_bit_cast(bf16x8, both), pf[kt][st], o[c2]);
                }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) {
                bf16x4 w4;
#pragma unroll
                for (int e = 0; e < 4; ++e) w4[e] = (bf16)Elem<E>::sat(o[c2][4 * jg + e]);
                *reinterpret_cast<bf16x4*>(orow + chunk * 64 + c2 * 32 + 8 * jg) = w4;
            }
        if (more) {
            if constexpr (PF == 2) commit_s(AttTag<PAR ^ (PF - 1)>{}, (chunk + 1) & 1, VRS);
            else commit((chunk + 1) & 1, VRS);
        }
        __syncthreads();
    };
    if constexpr (PF == 2) {
        for (int chunk = 0; chunk < nch; chunk += 2) { v_body(chunk, AttTag<0>{}); v_body(chunk + 1, AttTag<PF - 1>{}); }
    } else {
        for (int chunk = 0; chunk < nch; ++chunk) v_body(chunk, AttTag<0>{});
    }
}

template <int NKT, int NW, typename E, int PF = 1>
static int launch_attention_v2(const void* qkv, void* out, int B, int C, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 32 * NKT * HSIDM_ATT_VRS * 2;
    static PerDeviceOnce once;
    if (int rc = raise_lds_cap(once, &attention_v2_kernel<NKT, NW, E, PF>, lds)) return rc;
    dim3 grid(NKT / NW, B);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(attention_v2_kernel<NKT, NW, E, PF>), grid, dim3(64 * NW), lds, s, (const E*)qkv, (E*)out, C,
                       1.0f / sqrtf((float)C));
    return (int)hipGetLastError();
}

// ---- attention_v2, fp32 mode (fp32 storage; every operand as bf16 hi + lo, three MFMAs per product) ------------------
// The structure of attention_v2_kernel with the split done where the operand is produced: K and V chunks are split at the commit into
// a hi and a lo image in LDS (32-channel chunks: both images of both buffers are 96 KiB at 256 keys, one workgroup per CU), the Q
// fragments are split in registers as they arrive, the probabilities are split in the registers their scores occupied.  Replaces the
// score-panel kernel above for the UNet's shapes in the fp32 mode (6 launches x 775 us of a 66.7 ms step at 240 images).
template <int NKT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void attention_v2_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, float scale) {
    using bx8 = typename Elem<bf16>::x8;
    using bx4 = typename Elem<bf16>::x4;
    constexpr int N = 32 * NKT, T = 64 * NW, CH = 32, NV = N * (CH / 4) / T;
    // bf16 per key row: 80 B for the K image (ds_read_b128 fragments: an odd multiple of 16 B), 96 B for the V image (transposed reads:
    // the four key rows of a 16-lane group start 24 banks apart, eight banks each - conflict-free)
    constexpr int KRS = 40, VRS = 48;
    constexpr int IMG = N * VRS, BUFE = 2 * IMG;                       // [hi | lo]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* buf = reinterpret_cast<bf16*>(smem_raw);                     // [2][BUFE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 32 * NW + 32 * wave;
    const size_t row3 = (size_t)3 * C;
    const float* base = qkv + (size_t)b * N * row3;
    const int nch = C / CH;

    f32x4 hreg[NV];
    auto issue = [&](int chunk, int third) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * T;
            hreg[i] = *reinterpret_cast<const f32x4*>(base + (size_t)(v >> 3) * row3 + third * C + chunk * CH + (v & 7) * 4);
        }
    };
    auto commit = [&](int bi, int rs) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * T;
            bx4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = (bf16)hreg[i][e];
                lo[e] = (bf16)(hreg[i][e] - (float)hi[e]);
            }
            bf16* dst = buf + bi * BUFE + (v >> 3) * rs + (v & 7) * 4;
            *reinterpret_cast<bx4*>(dst) = hi;
            *reinterpret_cast<bx4*>(dst + IMG) = lo;
        }
    };
    auto split8 = [&](const f32x4& x, const f32x4& y, bx8& hi, bx8& lo) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (bf16)x[e];
            lo[e] = (bf16)(x[e] - (float)hi[e]);
            hi[4 + e] = (bf16)y[e];
            lo[4 + e] = (bf16)(y[e] - (float)hi[4 + e]);
        }
    };

    // ---- phase 1: S^T[key][query] = K Q^T ---------------------------------------------------------------------------------------
    const float* qrow = base + (size_t)(q0 + lr) * row3 + 8 * lh;
    f32x4 qraw[2][2];
    bx8 qh[2], ql[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        qraw[kk][0] = *reinterpret_cast<const f32x4*>(qrow + kk * 16);
        qraw[kk][1] = *reinterpret_cast<const f32x4*>(qrow + kk * 16 + 4);
    }
    issue(0, 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) split8(qraw[kk][0], qraw[kk][1], qh[kk], ql[kk]);
    commit(0, KRS);
    __syncthreads();
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int j = 0; j < 16; ++j) sc[kt][j] = 0.f;
    for (int chunk = 0; chunk < nch; ++chunk) {
        const bool more = chunk + 1 < nch;
        if (more) {
            issue(chunk + 1, 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                qraw[kk][0] = *reinterpret_cast<const f32x4*>(qrow + (chunk + 1) * CH + kk * 16);
                qraw[kk][1] = *reinterpret_cast<const f32x4*>(qrow + (chunk + 1) * CH + kk * 16 + 4);
            }
        }
        const bf16* kb = buf + (chunk & 1) * BUFE + lr * KRS + 8 * lh;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const bx8 ah = *reinterpret_cast<const bx8*>(kb + kt * 32 * KRS + kk * 16);
                const bx8 al = *reinterpret_cast<const bx8*>(kb + IMG + kt * 32 * KRS + kk * 16);
                sc[kt] = Elem<bf16>::mfma(al, qh[kk], sc[kt]);
                sc[kt] = Elem<bf16>::mfma(ah, ql[kk], sc[kt]);
                sc[kt] = Elem<bf16>::mfma(ah, qh[kk], sc[kt]);
            }
        if (more) {
            commit((chunk + 1) & 1, KRS);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) split8(qraw[kk][0], qraw[kk][1], qh[kk], ql[kk]);
        }
        __syncthreads();
    }

    // ---- exact softmax over the keys of this lane's query (fp32) -----------------------------------------------------------------
    issue(0, 2);
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int j = 0; j < 16; ++j) { sc[kt][j] *= scale; m = fmaxf(m, sc[kt][j]); }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int j = 0; j < 16; ++j) { sc[kt][j] = __expf(sc[kt][j] - m); sum += sc[kt][j]; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    bx8 ph[NKT][2], pl[NKT][2];                            // P^T fragments, hi and lo (attention_v2_kernel: k order of an accumulator tile)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = sc[kt][8 * st + j] * inv;
                ph[kt][st][j] = (bf16)pv;
                pl[kt][st][j] = (bf16)(pv - (float)ph[kt][st][j]);
            }
    commit(0, VRS);
    __syncthreads();

    // ---- phase 2: O^T[channel][query] = V^T P^T, 32 channels per chunk ----------------------------------------------------------
    const int gi = lane & 15;
    const int tr_lane = (4 * lh + (gi >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (gi & 3);
    float* orow = out + ((size_t)b * N + q0 + lr) * C + 4 * lh;
    for (int chunk = 0; chunk < nch; ++chunk) {
        const bool more = chunk + 1 < nch;
        if (more) issue(chunk + 1, 2);
        const bf16* vb = buf + (chunk & 1) * BUFE + tr_lane;
        f32x16 o;
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const bf16* a0 = vb + (kt * 32 + st * 16) * VRS;
                const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0));
                const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0 + 8 * VRS));
                const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0 + IMG));
                const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0 + IMG + 8 * VRS));
                const s16x8 vh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                const s16x8 vl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                o = Elem<bf16>::mfma(__builtin_bit_cast(bx8, vl), ph[kt][st], o);
                o = Elem<bf16>::mfma(__builtin_bit_cast(bx8, vh), pl[kt][st], o);
                o = Elem<bf16>::mfma(__builtin_bit_cast(bx8, vh), ph[kt][st], o);
            }
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
            const f32x4 w4 = {o[4 * jg], o[4 * jg + 1], o[4 * jg + 2], o[4 * jg + 3]};
            *reinterpret_cast<f32x4*>(orow + chunk * CH + 8 * jg) = w4;
        }
        if (more) commit((chunk + 1) & 1, VRS);
        __syncthreads();
    }
}

template <int NKT, int NW>
static int launch_attention_v2_f32(const void* qkv, void* out, int B, int C, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 2 * 32 * NKT * 48 * 2;
    static PerDeviceOnce once;
    if (int rc = raise_lds_cap(once, &attention_v2_f32_kernel<NKT, NW>, lds)) return rc;
    dim3 grid(NKT / NW, B);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(attention_v2_f32_kernel<NKT, NW>), grid, dim3(64 * NW), lds, s, (const float*)qkv, (float*)out, C,
                       1.0f / sqrtf((float)C));
    return (int)hipGetLastError();
}

}  // namespace hsidm

extern "C" int hsidm_attention(int prec, const void* qkv, void* out, int B, int N, int C, void* stream) {
    if (!qkv || !out || B <= 0 || N <= 0 || C <= 0 || (C & 31)) return HSIDM_E_BADARG;
    if (N > 1024) return HSIDM_E_UNSUPPORTED;
    // N = 256: ONE workgroup of 8 waves per image when the batch fills at least half of the CUs with such workgroups - K and V are then
    // read once per image instead of once per 128-query workgroup (-33 % of the launch's bytes, -8 % of its time at 240 images: the launch
    // is HBM-bound, 128 FLOP per byte; profiles/r06_attention/ab_forms.txt) - else two workgroups of 4 waves (more workgroups in flight).
    // (diagnostic A/B switches: HSIDM_ATTENTION_V1=1 the score-panel kernel, =2 the 4-wave form whatever the batch, =3 the 8-wave form
    // with the chunk stream requested ONE chunk ahead instead of two)
    const int att = hsidm::debug_get(hsidm::DBG_ATTENTION_V1);
    const bool one_wg = att != 2 && 2 * B >= hsidm::device_cus();
    if (prec == HSIDM_BF16 && (C & 63) == 0 && att != 1) {
        if (N == 256 && one_wg) return (att != 3 && (C & 127) == 0) ? hsidm::launch_attention_v2<8, 8, hsidm::bf16, 2>(qkv, out, B, C, (hipStream_t)stream)
                                                                  : hsidm::launch_attention_v2<8, 8, hsidm::bf16>(qkv, out, B, C, (hipStream_t)stream);
        if (N == 256) return hsidm::launch_attention_v2<8, 4, hsidm::bf16>(qkv, out, B, C, (hipStream_t)stream);
        if (N == 64) return hsidm::launch_attention_v2<2, 2, hsidm::bf16>(qkv, out, B, C, (hipStream_t)stream);
    }
    if (prec == HSIDM_F16 && (C & 63) == 0 && att != 1) {
        if (N == 256 && one_wg) return (att != 3 && (C & 127) == 0) ? hsidm::launch_attention_v2<8, 8, hsidm::f16, 2>(qkv, out, B, C, (hipStream_t)stream)
                                                                  : hsidm::launch_attention_v2<8, 8, hsidm::f16>(qkv, out, B, C, (hipStream_t)stream);
        if (N == 256) return hsidm::launch_attention_v2<8, 4, hsidm::f16>(qkv, out, B, C, (hipStream_t)stream);
        if (N == 64) return hsidm::launch_attention_v2<2, 2, hsidm::f16>(qkv, out, B, C, (hipStream_t)stream);
    }
    if (prec == HSIDM_F16) return hsidm::launch_attention<hsidm::f16, false, hsidm::f16>(qkv, out, B, N, C, (hipStream_t)stream);
    if (prec == HSIDM_BF16) return hsidm::launch_attention<hsidm::bf16, false>(qkv, out, B, N, C, (hipStream_t)stream);
    if (prec == HSIDM_F32X3 && att != 1) {
        if (N == 256) return hsidm::launch_attention_v2_f32<8, 4>(qkv, out, B, C, (hipStream_t)stream);
        if (N == 64) return hsidm::launch_attention_v2_f32<2, 2>(qkv, out, B, C, (hipStream_t)stream);
    }
    if (prec == HSIDM_F32X3) return hsidm::launch_attention<float, true>(qkv, out, B, N, C, (hipStream_t)stream);
    return HSIDM_E_BADARG;
}
