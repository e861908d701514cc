// Implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16).
//
//   D[pixel][cout] = sum_{tap, cin} T(x)[pixel + tap][cin] * W[tap][cin][cout]
//
// One workgroup (256 threads = 4 waves) owns BM = 128 output pixels (a TH x TW spatial tile of
// NI images) x BN output channels.  The K loop walks "steps" = (phase, cin-chunk of BK, tap):
//   * once per cin-chunk the (TH*S+2) x (TW*S+2) input HALO tile is fetched (NHWC, 16-B
//     vectors, coalesced along channels), the producer-side transform T (GroupNorm affine
//     [+ SiLU]) is applied in registers, and the bf16 result is written to LDS; all 9 taps then
//     read shifted windows of that one LDS tile, so every input element leaves HBM/L2 once per
//     (chunk, cout-slice) instead of 9 times and nothing normalised is ever materialised;
//   * once per step the BN x BK weight tile (pre-packed, one contiguous 16-B-vector stream) is
//     prefetched into registers while the previous step's MFMAs run, then committed to the
//     other LDS buffer (register-staged double buffering, one barrier per step).
// Phase 1 (optional) appends the 1x1 residual projection res_conv(x) to the same accumulators
// (reference unet.py:102-103,111), so the ResnetBlock tail is a single launch.
//
// Precision modes (template):
//   ActT = bf16, SPLIT = false : bf16 storage, bf16 operands, fp32 accumulate  ("bf16")
//   ActT = float, SPLIT = true : fp32 storage; each operand is split into bf16 hi + bf16 lo and
//                                hi*hi + hi*lo + lo*hi is accumulated in fp32 ("fp32x3",
//                                ~2^-17 relative per product; used for the fp32 parity gate)
//   ActT = f16, Op = f16, WLO   : fp16 storage and operands; the weights as fp16 hi + lo (two MFMAs per product): what the
//                                shapes the persistent kernels refuse run on in the fp16 mode ("fp16"; always with the low
//                                halves - this kernel is not what a benchmark step runs on)
#pragma once
#include "common.h"

namespace hsidm {

enum { XF_NONE = 0, XF_AFFINE = 1, XF_AFFINE_SILU = 2 };
enum { ACT_NONE = 0, ACT_LEAKY = 1 };

struct ConvPhase {
    const void* src0;      // NHWC [B][Hin][Win][C0]
    const void* src1;      // NHWC [B][Hin][Win][C1] or null: channel-concat (src0, src1)
    const float2* gn_ab;   // [B][C0+C1] (scale, shift) of the fused GroupNorm, or null
    int C0, C1;
    int transform;         // XF_*
    int nchunks;           // ceil((C0+C1)/BK)
    int ntaps;             // 9 or 1
};

struct ConvParams {
    ConvPhase ph[2];
    int nphase;
    const bf16* w_hi;      // packed [step][Cout_pad][BK]
    const bf16* w_lo;      // same layout, low halves (SPLIT only)
    const float* bias;     // [Cout] or null
    const float* film;     // [B][film_stride] (already offset to this layer) or null
    int film_stride;
    const void* res;       // residual, same layout/type as out, or null
    float res_scale;       // out = res_scale * act(acc + bias + film) + res
    void* out;             // NHWC ActT [B][Hout][Wout][Cout]  (or NCHW fp32 when OUT_NCHW)
    int B, Hin, Win, Hout, Wout, Cout, Cout_pad;
    int ups;               // 1: input is nearest-x2 upsampled on the fly (Hout = 2*Hin)
    int act;               // ACT_*
    int tiles_x, tiles_y;
    float2* stats;         // optional [B][tiles_per_img*SUBS][Cout] (sum, sumsq) slab, each entry written once, or null
};

template <typename ActT, bool SPLIT_, int BN_, int BK_, int TH_, int TW_, int NI_, int KS_, int S_, bool OUT_NCHW_, typename Op_ = bf16,
          bool WLO_ = false>
struct ConvCfg {
    using Act = ActT;
    using Op = Op_;                                      // 16-bit MFMA operand type
    static constexpr bool SPLIT = SPLIT_;                // activations AND weights as hi + lo (three MFMAs per product)
    static constexpr bool WLO = WLO_ || SPLIT_;          // weights as hi + lo
    static constexpr bool OUT_NCHW = OUT_NCHW_;
    static constexpr int BN = BN_, BK = BK_, TH = TH_, TW = TW_, NI = NI_, KS = KS_, S = S_;
    static constexpr int BM = TH * TW * NI;
    static_assert(BM == 128, "M tile is 128 pixels");
    static constexpr int WN = (BN >= 64) ? 2 : 1;        // waves along N
    static constexpr int WM = 4 / WN;                    // waves along M
    static constexpr int MR = BM / WM / 32;              // 32x32 MFMA tiles per wave along M
    static constexpr int NR = BN / WN / 32;
    static constexpr int HROWS = (KS == 3) ? (TH - 1) * S + 3 : TH;
    static constexpr int HCOLS = (KS == 3) ? (TW - 1) * S + 3 : TW;
    static constexpr int HPIX = HROWS * HCOLS;
    static constexpr int PSTR = BK + 8;                  // +16 B pad: conflict-free ds_read_b128
    static constexpr int WSTR = BK + 8;
    static constexpr int VPP = BK / 8;                   // 8-channel vectors per pixel
    static constexpr int HVEC = NI * HPIX * VPP;
    static constexpr int MAXHV = (HVEC + 255) / 256;
    static constexpr int WVEC = BN * VPP;
    static constexpr int MAXWV = (WVEC + 255) / 256;
    static constexpr int NPART = SPLIT ? 2 : 1, WPART = WLO ? 2 : 1;
    static constexpr int HALO_ELEMS = NI * HPIX * PSTR;
    static constexpr int WT_ELEMS = BN * WSTR;
    static constexpr size_t LDS_BYTES = (size_t)(NPART * HALO_ELEMS + 2 * WPART * WT_ELEMS) * 2;
    // statistics slab: entries per (image, spatial tile) = one per wave row that covers that image
    static constexpr int SUBS = (NI == 1) ? WM : (WM >= 2 ? WM / 2 : 1);
};

template <typename ActT> struct RawVec;     // 8 activations as loaded from memory
template <> struct RawVec<bf16> { u32x4 a; };
template <> struct RawVec<f16> { u32x4 a; };
template <> struct RawVec<float> { u32x4 a, b; };

template <typename ActT>
__device__ __forceinline__ void raw_load(RawVec<ActT>& r, const ActT* p);
template <>
__device__ __forceinline__ void raw_load<bf16>(RawVec<bf16>& r, const bf16* p) {
    r.a = *reinterpret_cast<const u32x4*>(p);
}
template <>
__device__ __forceinline__ void raw_load<f16>(RawVec<f16>& r, const f16* p) {
    r.a = *reinterpret_cast<const u32x4*>(p);
}
template <>
__device__ __forceinline__ void raw_load<float>(RawVec<float>& r, const float* p) {
    r.a = *reinterpret_cast<const u32x4*>(p);
    r.b = *reinterpret_cast<const u32x4*>(p + 4);
}
template <typename ActT>
__device__ __forceinline__ void raw_zero(RawVec<ActT>& r);
template <>
__device__ __forceinline__ void raw_zero<bf16>(RawVec<bf16>& r) { r.a = u32x4{0, 0, 0, 0}; }
template <>
__device__ __forceinline__ void raw_zero<f16>(RawVec<f16>& r) { r.a = u32x4{0, 0, 0, 0}; }
template <>
__device__ __forceinline__ void raw_zero<float>(RawVec<float>& r) { r.a = u32x4{0, 0, 0, 0}; r.b = r.a; }

__device__ __forceinline__ void raw_unpack(const RawVec<bf16>& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(r.a[i] << 16);
        v[2 * i + 1] = __uint_as_float(r.a[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void raw_unpack(const RawVec<f16>& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = Elem<f16>::lo(r.a[i]);
        v[2 * i + 1] = Elem<f16>::hi(r.a[i]);
    }
}
__device__ __forceinline__ void raw_unpack(const RawVec<float>& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = __uint_as_float(r.a[i]); v[4 + i] = __uint_as_float(r.b[i]); }
}

template <typename C>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    using ActT = typename C::Act;
    constexpr int BN = C::BN, BK = C::BK, TH = C::TH, TW = C::TW, NI = C::NI, KS = C::KS, S = C::S;
    constexpr int MR = C::MR, NR = C::NR, PSTR = C::PSTR, WSTR = C::WSTR, VPP = C::VPP;
    constexpr int HPIX = C::HPIX, HCOLS = C::HCOLS;
    constexpr bool SPLIT = C::SPLIT, WLO = C::WLO;
    using Op = typename C::Op;
    using bf16 = Op;                                 // (the body below names its operand type bf16; Op = bf16 | fp16)
    using bf16x8 = typename Elem<Op>::x8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* halo_hi = reinterpret_cast<bf16*>(smem_raw);
    bf16* halo_lo = halo_hi + (SPLIT ? C::HALO_ELEMS : 0);
    bf16* wt_hi = halo_hi + C::NPART * C::HALO_ELEMS;            // [2][WT_ELEMS]
    bf16* wt_lo = wt_hi + (WLO ? 2 * C::WT_ELEMS : 0);           // [2][WT_ELEMS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- workgroup -> (image group, spatial tile, cout slice) -----------------------------
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int tile = blockIdx.x;
    const int tgrp = tile / tiles_per_img;
    const int trem = tile - tgrp * tiles_per_img;
    const int oy0 = (trem / p.tiles_x) * TH;
    const int ox0 = (trem % p.tiles_x) * TW;
    const int b0 = tgrp * NI;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread halo vector descriptors (independent of chunk / phase) ------------------
    int hv_pix[C::MAXHV];      // source pixel index (b*Hin + iy)*Win + ix, or -1 when zero-padded
    int hv_lds[C::MAXHV];      // destination element offset in the halo tile, or -1 when unused
    const int cv = tid % VPP;  // this thread's channel-vector slot (constant: 256 % VPP == 0)
#pragma unroll
    for (int i = 0; i < C::MAXHV; ++i) {
        const int v = tid + i * 256;
        hv_pix[i] = -1;
        hv_lds[i] = -1;
        if (v < C::HVEC) {
            const int hp = v / VPP;
            const int img = hp / HPIX;
            const int r = hp - img * HPIX;
            const int hy = r / HCOLS, hx = r - hy * HCOLS;
            hv_lds[i] = hp * PSTR + cv * 8;
            const int b = b0 + img;
            int iy, ix;
            bool ok = b < p.B;
            if (KS == 3) {
                iy = oy0 * S + hy - 1;
                ix = ox0 * S + hx - 1;
                if (p.ups) {   // coordinates are in the virtual upsampled image
                    ok = ok && iy >= 0 && ix >= 0 && iy < 2 * p.Hin && ix < 2 * p.Win;
                    iy >>= 1;
                    ix >>= 1;
                } else {
                    ok = ok && iy >= 0 && ix >= 0 && iy < p.Hin && ix < p.Win;
                }
            } else {
                iy = oy0 + hy;
                ix = ox0 + hx;
                ok = ok && iy < p.Hin && ix < p.Win;
            }
            if (ok) hv_pix[i] = (b * p.Hin + iy) * p.Win + ix;
        }
    }

    // ---- MFMA fragment base offsets ----------------------------------------------------------
    int abase[MR], bbase[NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int pm = wm * (C::BM / C::WM) + mr * 32 + lr;
        const int img = pm / (TH * TW);
        const int q = pm - img * (TH * TW);
        const int ty = q / TW, tx = q - ty * TW;
        abase[mr] = (img * HPIX + ty * S * HCOLS + tx * S) * PSTR + 8 * lh;
    }
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) bbase[nr] = (wn * (BN / C::WN) + nr * 32 + lr) * WSTR + 8 * lh;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mr][nr][j] = 0.f;

    RawVec<ActT> hreg[C::MAXHV];
    unsigned hvalid = 0;
    u32x4 wreg_hi[C::MAXWV], wreg_lo[WLO ? C::MAXWV : 1];

    // ---- staging helpers ---------------------------------------------------------------------------
    auto halo_issue = [&](const ConvPhase& ph, int chunk) {
        const int c = chunk * BK + cv * 8;        // channel in the virtual concat
        const ActT* src;
        int cs, cl;
        if (c < ph.C0) { src = (const ActT*)ph.src0; cs = ph.C0; cl = c; }
        else           { src = (const ActT*)ph.src1; cs = ph.C1; cl = c - ph.C0; }
        const bool cok = c < ph.C0 + ph.C1;
        hvalid = 0;
#pragma unroll
        for (int i = 0; i < C::MAXHV; ++i) {
            if (hv_pix[i] >= 0 && cok) {
                raw_load<ActT>(hreg[i], src + (size_t)hv_pix[i] * cs + cl);
                hvalid |= 1u << i;
            } else {
                raw_zero<ActT>(hreg[i]);
            }
        }
    };
    auto halo_commit = [&](const ConvPhase& ph, int chunk) {
        const int c = chunk * BK + cv * 8;
        const int ctot = ph.C0 + ph.C1;
        float sc[8], sh[8];
        const bool xf = ph.transform != XF_NONE;
        if (NI == 1 && xf && c < ctot) {
            const float2* ab = ph.gn_ab + (size_t)b0 * ctot + c;
#pragma unroll
            for (int k = 0; k < 8; ++k) { float2 t = ab[k]; sc[k] = t.x; sh[k] = t.y; }
        }
#pragma unroll
        for (int i = 0; i < C::MAXHV; ++i) {
            if (hv_lds[i] < 0) continue;
            float v[8];
            raw_unpack(hreg[i], v);
            if ((hvalid >> i) & 1u) {
                if (xf) {
                    if (NI > 1) {
                        const int img = (hv_lds[i] / PSTR) / HPIX;
                        const float2* ab = ph.gn_ab + (size_t)(b0 + img) * ctot + c;
#pragma unroll
                        for (int k = 0; k < 8; ++k) { float2 t = ab[k]; sc[k] = t.x; sh[k] = t.y; }
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float y = fmaf(v[k], sc[k], sh[k]);
                        v[k] = (ph.transform == XF_AFFINE_SILU) ? silu(y) : y;
                    }
                }
            }
            bf16x8 hi, lo;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hi[k] = (bf16)v[k];
                if (SPLIT) lo[k] = (bf16)(v[k] - (float)hi[k]);
            }
            *reinterpret_cast<bf16x8*>(halo_hi + hv_lds[i]) = hi;
            if (SPLIT) *reinterpret_cast<bf16x8*>(halo_lo + hv_lds[i]) = lo;
        }
    };
    auto w_issue = [&](int step) {
#pragma unroll
        for (int i = 0; i < C::MAXWV; ++i) {
            const int v = tid + i * 256;
            if (C::WVEC % 256 == 0 || v < C::WVEC) {
                const int n = v / VPP;
                const size_t off = ((size_t)step * p.Cout_pad + n0 + n) * BK + (v % VPP) * 8;
                wreg_hi[i] = *reinterpret_cast<const u32x4*>(p.w_hi + off);      // (2-byte elements of either operand type)
                if (WLO) wreg_lo[i] = *reinterpret_cast<const u32x4*>(p.w_lo + off);
            }
        }
    };
    auto w_commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::MAXWV; ++i) {
            const int v = tid + i * 256;
            if (C::WVEC % 256 == 0 || v < C::WVEC) {
                const int o = buf * C::WT_ELEMS + (v / VPP) * WSTR + (v % VPP) * 8;
                *reinterpret_cast<u32x4*>(wt_hi + o) = wreg_hi[i];
                if (WLO) *reinterpret_cast<u32x4*>(wt_lo + o) = wreg_lo[i];
            }
        }
    };
    auto mfma_step = [&](int buf, int aoff) {
        const bf16* wh = wt_hi + buf * C::WT_ELEMS;
        const bf16* wl = wt_lo + buf * C::WT_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 ah[MR], al[SPLIT ? MR : 1], bh[NR], bl[WLO ? NR : 1];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                ah[mr] = *reinterpret_cast<const bf16x8*>(halo_hi + abase[mr] + aoff + kk * 16);
                if (SPLIT) al[mr] = *reinterpret_cast<const bf16x8*>(halo_lo + abase[mr] + aoff + kk * 16);
            }
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                bh[nr] = *reinterpret_cast<const bf16x8*>(wh + bbase[nr] + kk * 16);
                if (WLO) bl[nr] = *reinterpret_cast<const bf16x8*>(wl + bbase[nr] + kk * 16);
            }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    // small terms first
                    if (SPLIT) acc[mr][nr] = Elem<Op>::mfma(al[SPLIT ? mr : 0], bh[nr], acc[mr][nr]);
                    if (WLO) acc[mr][nr] = Elem<Op>::mfma(ah[mr], bl[WLO ? nr : 0], acc[mr][nr]);
                    acc[mr][nr] = Elem<Op>::mfma(ah[mr], bh[nr], acc[mr][nr]);
                }
        }
    };

    // ---- main loop over (phase, chunk) x taps ---------------------------------------------------
    // Software pipeline: the next chunk's halo vectors and the next step's weight vectors are in
    // flight (global -> registers) while the current step's MFMAs run.
    const int nch0 = p.ph[0].nchunks;
    const int nch = nch0 + (p.nphase > 1 ? p.ph[1].nchunks : 0);
    int step = 0;
    int wbuf = 0;
    w_issue(0);
    halo_issue(p.ph[0], 0);
    for (int gi = 0; gi < nch; ++gi) {
        const ConvPhase& ph = (gi < nch0) ? p.ph[0] : p.ph[1];
        const int chunk = (gi < nch0) ? gi : gi - nch0;
        const bool center_only = (KS == 3) && (ph.ntaps == 1);   // fused 1x1 residual projection
        __syncthreads();                     // every wave finished reading the previous halo tile
        halo_commit(ph, chunk);
        if (gi + 1 < nch) halo_issue((gi + 1 < nch0) ? p.ph[0] : p.ph[1], (gi + 1 < nch0) ? gi + 1 : gi + 1 - nch0);
        for (int tap = 0; tap < ph.ntaps; ++tap, ++step) {
            w_commit(wbuf);
            __syncthreads();                 // halo + this step's weights visible
            const bool last = (gi == nch - 1) && (tap == ph.ntaps - 1);
            if (!last) w_issue(step + 1);
            int aoff = 0;
            if (KS == 3) {
                const int dy = center_only ? 1 : tap / 3;
                const int dx = center_only ? 1 : tap - 3 * (tap / 3);
                aoff = (dy * HCOLS + dx) * PSTR;
            }
            mfma_step(wbuf, aoff);
            wbuf ^= 1;
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    ActT* out = reinterpret_cast<ActT*>(p.out);
    const ActT* res = reinterpret_cast<const ActT*>(p.res);
    const int trem_e = trem;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int n = n0 + wn * (BN / C::WN) + nr * 32 + lr;
        const bool nok = n < p.Cout;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
        float s1[NI], s2[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) s1[q] = s2[q] = 0.f;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int pbase = wm * (C::BM / C::WM) + mr * 32;
            const int img = pbase / (TH * TW);             // a 32-row MFMA tile never straddles two images
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
                const int q = pbase + row - img * (TH * TW);
                const int ty = q / TW, tx = q - ty * TW;
                const int b = b0 + img, oy = oy0 + ty, ox = ox0 + tx;
                if (!(nok && b < p.B && oy < p.Hout && ox < p.Wout)) continue;
                float v = acc[mr][nr][j] + bias;
                if (p.film) v += p.film[(size_t)b * p.film_stride + n];
                if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
                if (C::OUT_NCHW) {
                    const size_t o = (((size_t)b * p.Cout + n) * p.Hout + oy) * p.Wout + ox;
                    float* of = reinterpret_cast<float*>(p.out);
                    if (p.res) v = p.res_scale * v + reinterpret_cast<const float*>(p.res)[o];
                    of[o] = v;
                } else {
                    const size_t o = (((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + n;
                    if (p.res) v = p.res_scale * v + to_f32<ActT>(res[o]);
                    const ActT st = from_f32<ActT>(v);
                    out[o] = st;
                    const float sv = to_f32<ActT>(st);      // statistics of the tensor exactly as stored
                    s1[NI == 1 ? 0 : img] += sv;
                    s2[NI == 1 ? 0 : img] += sv * sv;
                }
            }
        }
        if (!C::OUT_NCHW && p.stats) {
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const float a = s1[q] + __shfl_xor(s1[q], 32, 64);
                const float d = s2[q] + __shfl_xor(s2[q], 32, 64);
                int img, sub;
                if (NI == 1) { img = 0; sub = wm; }
                else if (C::WM == 1) { img = q; sub = 0; }
                else { img = (wm * (C::BM / C::WM)) / (TH * TW); sub = wm % (C::WM / 2 > 0 ? C::WM / 2 : 1); if (q != img) continue; }
                const int b = b0 + img;
                if (lh == 0 && nok && b < p.B)
                    p.stats[((size_t)b * (tiles_per_img * C::SUBS) + trem_e * C::SUBS + sub) * p.Cout + n] = make_float2(a, d);
            }
        }
    }
}

}  // namespace hsidm
