// conv_sk: split-K form of the bf16 3x3 stride-1 convolution for launches with FEW pixel tiles and a LONG contraction - the 8x8 and
// 16x16 levels of the UNet (Cin 512...1024, Cout 512) at small batches: one CAVE image (5 latents), the 8-GPU shard of configs[3]
// (40 latents), a training step (4 latents, forward and input gradient).
//
// Why: the persistent kernel (conv_v2.h) gives such a launch tiles x slices work items - 20 at 5 latents on the 8x8 level, 80 at 40 -
// for 512 workgroup slots, and every item walks the whole contraction: 9 taps x 16 chunks in sequence.  Measured at 5 latents
// (profiles/r02_small_batch/conv_bench_b5.txt): 8x8 level 1024 -> 512 in 78.6 us = 38 TFLOP/s, the two deep levels 1.0 ms of a
// 2.83 ms step.  The contraction is the only axis left to parallelise.
//
// Here a workgroup owns (64 pixels of one image) x (128 couts) x (a RANGE of 64-channel chunks, all 9 taps): grid = tiles x slices x
// parts.  Partial sums go to an fp32 workspace [part][pixel][cout] with plain stores, and conv_sk_finish adds the parts in order
// and applies the epilogue the fused kernel has (bias, FiLM, residual, bf16 store, GroupNorm statistics of the stored tensor):
// deterministic, no atomics.  Operand formats are the persistent kernel's: NHWC activations (two-pointer concat, GroupNorm + SiLU
// applied while staging), weights in the register-streaming order w_v2 (one contiguous 1 KiB per wave-load of an MFMA B fragment).
#include "conv_v2.h"
#include "../../include/hsidm.h"

namespace hsidm {

struct ConvSkParams {
    const bf16* src0; const bf16* src1;
    const float2* gn_ab;            // fp32 (scale, shift) pairs [B][C0+C1], or null
    const bf16* w;                  // w_v2 layout
    float* partial;                 // [parts][B*H*W][Cout]
    int C0, C1, nchunks, cpp;       // chunks of 64 channels; chunks per part
    // fused 1x1 projection of a second input (ResnetBlock.res_conv, reference unet.py:102-103,110): pchunks more chunks of ONE tap
    // each, untransformed; their weight steps follow the 9 * nchunks steps of the 3x3 kernel
    const bf16* psrc0; const bf16* psrc1;
    int PC0, PC1, pchunks;
    int B, H, W, Hin, Win, Cout, Cout_pad;          // H, W: output map; Hin, Win: input map (= H, W unless stride 2)
    int tiles_x, tiles_y;
    int silu;
};

// S = 1: 10 x 10 halo of an 8 x 8 output tile.  S = 2 (Downsample, reference unet.py:68-74: 3x3, stride 2, pad 1): 17 x 17 input
// pixels, output (r, c) reads input (2r + ky, 2c + kx) of the tile; the weights are the parity-plane layout the persistent kernel
// uses for these layers (include/hsidm.h: stride = 2 with w_v2), addressed by the plane / window tap of each 3x3 tap.
template <int S>
struct SkGeo {
    static constexpr int TH = 8, TW = 8, HR = S * TH + 3 - S, HC = S * TW + 3 - S, HPIX = HR * HC, PSTR = 72, VPP = 8;
    static constexpr int HVEC = HPIX * VPP, MAXHV = (HVEC + 255) / 256;   // S = 1: 800 vectors, 4 per thread; S = 2: 2312, 10 per thread
};

template <int S, typename E>
__global__ __launch_bounds__(256, 2) void conv_sk_kernel(const ConvSkParams p) {
    using G = SkGeo<S>;
    using EL = Elem<E>;
    using x8 = typename EL::x8;
    constexpr int TH = G::TH, TW = G::TW, HC = G::HC, HPIX = G::HPIX, PSTR = G::PSTR, MAXHV = G::MAXHV;
    __shared__ __attribute__((aligned(16))) E halo[HPIX * PSTR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int tile = blockIdx.x, slice = blockIdx.y, part = blockIdx.z;
    const int tpi = p.tiles_x * p.tiles_y;
    const int b = tile / tpi, tr = tile - b * tpi;
    const int ty0 = (tr / p.tiles_x) * TH, tx0 = (tr % p.tiles_x) * TW;
    const int c_begin = part * p.cpp, c_end = min(p.nchunks + p.pchunks, c_begin + p.cpp);

    // weight fragments of this wave's 32 couts: [step][Cout_pad/32][kk][lane][8]
    const int nsw = p.Cout_pad >> 5;
    const E* wlane = reinterpret_cast<const E*>(p.w) + ((size_t)(slice * 4 + wave) * 4 * 64 + lane) * 8;
    const size_t wstep = (size_t)nsw * 4 * 64 * 8;

    // A fragment base of the two 32-pixel MFMA tiles (tile rows 4 mt .. 4 mt + 3): lane = (row lr / 8, column lr % 8), k half lh
    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) abase[mt] = (S * (4 * mt + (lr >> 3)) * HC + S * (lr & 7)) * PSTR + 8 * lh;

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mt][j] = 0.f;

    const int cv = tid & 7;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        // the chunk's 36 weight fragments (9 taps x 4 k-slices, 1 KiB per wave-load) are requested first: their L2 latency hides
        // behind the staging of the halo tile (a one-tap-ahead ring was latency-bound: 8 MFMAs per tap against ~0.7 us per fetch)
        const bool proj = chunk >= p.nchunks;                    // workgroup-uniform
        const int lc = proj ? chunk - p.nchunks : chunk;         // chunk within its phase
        const E* wc = wlane + (size_t)(proj ? p.nchunks * 9 + lc : lc * 9) * wstep;
        x8 wr[9][4];
        if (!proj) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // S = 2: 3x3 tap (dy, dx) lives in plane (dy != 1, dx != 1) at window tap (dy == 1 ? 1 : dy / 2, dx likewise);
                // step = (plane * nchunks + chunk) * 4 + window tap
                const int dy = tap / 3, dx = tap % 3;
                const int plane = 2 * (dy != 1) + (dx != 1), wt = 2 * (dy == 1 ? 1 : dy / 2) + (dx == 1 ? 1 : dx / 2);
                const E* wt_p = S == 1 ? wc + (size_t)tap * wstep : wlane + ((size_t)(plane * p.nchunks + lc) * 4 + wt) * wstep;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wr[tap][kk] = *reinterpret_cast<const x8*>(wt_p + kk * 64 * 8);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wr[0][kk] = *reinterpret_cast<const x8*>(wc + kk * 64 * 8);
        }
        // ---- stage the 10 x 10 halo of this chunk's 64 channels (GroupNorm + SiLU on the way; zero padding AFTER the activation) ----
        const int ctot = proj ? p.PC0 + p.PC1 : p.C0 + p.C1;
        const int c = lc * 64 + cv * 8;
        const bool cok = c < ctot;
        const int cc = cok ? c : 0;
        const E* src;
        int cs, cl;
        if (!proj) { if (cc < p.C0) { src = reinterpret_cast<const E*>(p.src0); cs = p.C0; cl = cc; } else { src = reinterpret_cast<const E*>(p.src1); cs = p.C1; cl = cc - p.C0; } }
        else       { if (cc < p.PC0) { src = reinterpret_cast<const E*>(p.psrc0); cs = p.PC0; cl = cc; } else { src = reinterpret_cast<const E*>(p.psrc1); cs = p.PC1; cl = cc - p.PC0; } }
        const float2* gn = proj ? nullptr : p.gn_ab;
        float sc[8], sh[8];
        if (gn) {
            const f32x4* q = reinterpret_cast<const f32x4*>(gn + (size_t)b * ctot + cc);
            f32x4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = q[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) { sc[2 * k] = r[k][0]; sh[2 * k] = r[k][1]; sc[2 * k + 1] = r[k][2]; sh[2 * k + 1] = r[k][3]; }
        }
        u32x4 raw[MAXHV];
        bool okv[MAXHV];
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) {
            const int hp = min((tid >> 3) + i * 32, HPIX - 1);
            const int hy = hp / HC, hx = hp - hy * HC;
            const int iy = S * ty0 + hy - 1, ix = S * tx0 + hx - 1;
            okv[i] = cok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const int cy = min(max(iy, 0), p.Hin - 1), cx = min(max(ix, 0), p.Win - 1);
            raw[i] = *reinterpret_cast<const u32x4*>(src + ((size_t)(b * p.Hin + cy) * p.Win + cx) * cs + cl);
        }
        __syncthreads();                                         // the previous chunk's readers are done
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) {
            const int hp = (tid >> 3) + i * 32;
            u32x4 o4 = raw[i];
            if (gn) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] = EL::lo(o4[k]); v[2 * k + 1] = EL::hi(o4[k]); }
                x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float u = fmaf(v[k], sc[k], sh[k]);
                    o[k] = (E)EL::sat(p.silu ? silu_fast(u) : u);
                }
                o4 = __builtin_bit_cast(u32x4, o);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) o4[k] = okv[i] ? o4[k] : 0u;
            if (hp < HPIX) *reinterpret_cast<u32x4*>(halo + hp * PSTR + cv * 8) = o4;
        }
        __syncthreads();
        // ---- 9 taps x 4 k-slices (projection chunk: the centre tap only) ----
        if (!proj) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * HC + (tap % 3)) * PSTR;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const x8 a = *reinterpret_cast<const x8*>(halo + abase[mt] + toff + kk * 16);
                        acc[mt] = EL::mfma(a, wr[tap][kk], acc[mt]);
                    }
                }
            }
        } else {
            const int toff = (HC + 1) * PSTR;                   // (projections only come with S = 1)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const x8 a = *reinterpret_cast<const x8*>(halo + abase[mt] + toff + kk * 16);
                    acc[mt] = EL::mfma(a, wr[0][kk], acc[mt]);
                }
            }
        }
    }
    // ---- partial sums: register j of a lane = tile pixel (j&3) + 8(j>>2) + 4 lh of its 32-pixel MFMA tile, column = cout ----
    const int n = slice * 128 + wave * 32 + lr;
    if (n < p.Cout) {
        float* dst = p.partial + (size_t)part * p.B * p.H * p.W * p.Cout;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int pr = (j & 3) + 8 * (j >> 2) + 4 * lh;
                const int y = ty0 + 4 * mt + (pr >> 3), x = tx0 + (pr & 7);
                if (y < p.H && x < p.W) dst[((size_t)(b * p.H + y) * p.W + x) * p.Cout + n] = acc[mt][j];
            }
    }
}

// out = res_scale * (sum of the parts + bias + FiLM) + res, rounded to bf16; statistics slab [B][HW/64][Cout] of what was stored.
// grid (HW / 64, B, Cout / 32): a workgroup owns 64 pixels x 32 channels; thread = (channel vector of 8, pixel).
template <typename E>
__global__ __launch_bounds__(256) void conv_sk_finish_kernel(const float* __restrict__ partial, int nparts, const float* __restrict__ bias,
                                                             const float* __restrict__ film, int film_stride, const E* __restrict__ res,
                                                             float res_scale, E* __restrict__ out, float2* __restrict__ stats, int B, int HW,
                                                             int Cout) {
    using x8 = typename Elem<E>::x8;
    __shared__ float red[64][32][2];
    const int grp = blockIdx.x, b = blockIdx.y, cg = blockIdx.z;
    const int t = threadIdx.x, cvi = t & 3, prow = t >> 2;
    const int c = cg * 32 + cvi * 8;
    const size_t plane = (size_t)B * HW * Cout;
    float s1[8], s2[8], add[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s1[k] = s2[k] = 0.f;
        add[k] = (bias ? bias[c + k] : 0.f) + (film ? film[(size_t)b * film_stride + c + k] : 0.f);
    }
    {
        const int pix = grp * 64 + prow;
        const size_t off = ((size_t)b * HW + pix) * Cout + c;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        int q = 0;
        for (; q + 3 < nparts; q += 4) {                         // four parts in flight, added in order
            f32x4 a0[4], a1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0[j] = *reinterpret_cast<const f32x4*>(partial + (size_t)(q + j) * plane + off);
                a1[j] = *reinterpret_cast<const f32x4*>(partial + (size_t)(q + j) * plane + off + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] += a0[j][k]; v[4 + k] += a1[j][k]; }
        }
        for (; q < nparts; ++q) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(partial + (size_t)q * plane + off);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(partial + (size_t)q * plane + off + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] += a0[k]; v[4 + k] += a1[k]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = res_scale * (v[k] + add[k]);
        if (res) {
            const x8 r = *reinterpret_cast<const x8*>(res + off);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += (float)r[k];
        }
        x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o[k] = (E)Elem<E>::sat(v[k]);
            s1[k] += v[k];
            s2[k] = fmaf(v[k], v[k], s2[k]);
        }
        *reinterpret_cast<x8*>(out + off) = o;
    }
    if (stats) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[prow][cvi * 8 + k][0] = s1[k]; red[prow][cvi * 8 + k][1] = s2[k]; }
        __syncthreads();
        if (t < 32) {
            float a = 0.f, d = 0.f;
            for (int r = 0; r < 64; ++r) { a += red[r][t][0]; d += red[r][t][1]; }
            stats[((size_t)b * gridDim.x + grp) * Cout + cg * 32 + t] = make_float2(a, d);
        }
    }
}

// chunks per part: one round of co-resident workgroups (two per CU) over the launch, at least one chunk each (measured: 640
// workgroups of two chunks = two rounds took 33.7 us on the 16x16 level at 5 latents; a chunk costs ~8 us of mostly latency)
static int sk_cpp(int tiles, int slices, int nchunks) {
    const int want = 2 * device_cus();
    int parts = want / (tiles * slices);
    if (parts > nchunks) parts = nchunks;
    if (parts < 1) parts = 1;
    return (nchunks + parts - 1) / parts;
}

int conv_sk_parts(int B, int H, int W, int Cout, int nchunks) {
    const int tiles = B * (H / 8) * (W / 8), slices = Cout / 128;
    const int cpp = sk_cpp(tiles, slices, nchunks);
    return (nchunks + cpp - 1) / cpp;
}

int conv_sk_run(const bf16* src0, const bf16* src1, int C0, int C1, const float2* gn_ab, int silu, const bf16* w, const float* bias,
                const float* film, int film_stride, const bf16* res, float res_scale, bf16* out, float2* stats, int B, int H, int W,
                int Cout, int Cout_pad, int nchunks, const bf16* psrc0, const bf16* psrc1, int PC0, int PC1, int stride, float* workspace,
                int elem, hipStream_t s) {
    ConvSkParams p;
    p.src0 = src0; p.src1 = src1; p.gn_ab = gn_ab; p.w = w; p.partial = workspace;
    p.C0 = C0; p.C1 = C1; p.nchunks = nchunks;
    p.psrc0 = psrc0; p.psrc1 = psrc1; p.PC0 = PC0; p.PC1 = PC1; p.pchunks = (PC0 + PC1 + 63) / 64;
    p.B = B; p.H = H; p.W = W; p.Hin = stride * H; p.Win = stride * W; p.Cout = Cout; p.Cout_pad = Cout_pad;
    p.tiles_x = W / 8; p.tiles_y = H / 8;
    p.silu = silu;
    const int tiles = B * p.tiles_x * p.tiles_y, slices = Cout / 128;
    const int allchunks = nchunks + p.pchunks;
    p.cpp = sk_cpp(tiles, slices, allchunks);
    const int parts = (allchunks + p.cpp - 1) / p.cpp;
    const dim3 g(tiles, slices, parts), gf(H * W / 64, B, Cout / 32);
    if (elem == 0) {
        if (stride == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_sk_kernel<2, bf16>), g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_sk_kernel<1, bf16>), g, dim3(256), 0, s, p);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_sk_finish_kernel<bf16>), gf, dim3(256), 0, s, (const float*)workspace, parts, bias, film,
                           film_stride, res, res_scale, out, stats, B, H * W, Cout);
    } else {
        if (stride == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_sk_kernel<2, f16>), g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_sk_kernel<1, f16>), g, dim3(256), 0, s, p);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_sk_finish_kernel<f16>), gf, dim3(256), 0, s, (const float*)workspace, parts, bias, film,
                           film_stride, reinterpret_cast<const f16*>(res), res_scale, reinterpret_cast<f16*>(out), stats, B, H * W, Cout);
    }
    return (int)hipGetLastError();
}

}  // namespace hsidm
