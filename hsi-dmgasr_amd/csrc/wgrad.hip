// Weight gradient of the path's convolutions (the reference gets it from autograd in model/model.py:55):
//   dw[co][ci][ky][kx] = sum over (b, y, x) of dy[b][y][x][co] * a[b][s*y + ky - 1][s*x + kx - 1][ci]
// as an implicit GEMM whose contraction axis is the PIXEL axis: D[co][ci] (per tap) += dY^T[co][pixel] * A[pixel (+tap)][ci].
// Both operands are NHWC (channels contiguous), i.e. the contraction index is the strided one for both.  bf16 mode: a pixel tile
// of dY and the halo tile of A are staged row-major in LDS ([pixel][64 channels], 192-byte rows) and BOTH MFMA operands are
// fetched with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-major: conflict-free at this
// pitch), so no transposed copy of either tensor is ever written.  fp32 mode: the fp32 tensors are split into bf16 hi + lo while
// they are staged (four LDS tiles of the bf16 kind) and every product is three MFMAs, hi*hi + hi*lo + lo*hi - the forward
// kernels' fp32 mode (conv_igemm.h), ~2^-17 relative per product.  (Round 2 used the exact fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 here: one element per lane at 1/16 of the bf16 rate, 18 of the fp32-mode step's 28 ms.)
// Work decomposition: workgroup = (64 couts x 64 cins x all taps) x a contiguous range of pixel tiles (split K); a wave owns a
// 32 x 32 corner for all taps (9 accumulator tiles); partial sums go to a workspace [split][tap][cout][cin] with plain stores and
// a second kernel adds the splits in order into the PyTorch layout [cout][cin][ky][kx]: deterministic, no atomics.
// Bias gradient (sum of dy over the pixels) on request: the workgroups of cin tile 0 carry one more accumulator tile whose B operand
// is all ones - D[co][*] += dY^T[co][pixel] * 1 - fed by the dY fragments they fetch anyway: one MFMA per k-group, no extra LDS
// read and no reduction code; its column 0 goes to the workspace like the other partial tiles.
#include "common.h"
#include "../../include/hsidm.h"
#include <type_traits>

namespace hsidm {

struct WgradParams {
    const void* a0; const void* a1; const void* dy;
    float* ws;
    float* bias_ws;                    // partial sums of dy over the pixels (the bias gradient), or null:
    int bias_images;                   //   0: [nsplit][Cout_pad];  B: per image, [nsplit][B][Cout_pad] (FiLM's gradient needs them apart)
    int C0, C1, Cin, Cout;             // tensor channel counts (multiples of 8)
    int B, Hin, Win, Hout, Wout;
    int tiles_x, tiles_y, ksteps, nsplit;
    int cin_tiles, Cin_pad, Cout_pad;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// MODE 0: stride 1 (3x3 pad 1, or 1x1);  1: the input is read through a nearest x2 upsample;  2: stride 2 (3x3 pad 1)
template <typename T, int TH, int TW, int NT, int MODE>
struct WgCfg {
    static constexpr bool F32 = std::is_same<T, float>::value;
    static constexpr int EPV = F32 ? 4 : 8;                  // elements per 16-byte vector
    static constexpr int VPP = 64 / EPV;                     // vectors per pixel (64 channels)
    static constexpr int PITCH = 96;                         // bf16 elements per LDS pixel row (fp32 mode: hi and lo tiles of this kind)
    static constexpr int NPART = F32 ? 2 : 1;
    static constexpr int NPIX = TH * TW;
    static constexpr int HH = NT == 1 ? TH : (MODE == 2 ? 2 * TH + 1 : TH + 2);
    static constexpr int HWD = NT == 1 ? TW : (MODE == 2 ? 2 * TW + 1 : TW + 2);
    static constexpr int NHALO = HH * HWD;
    static constexpr int NV_D = (NPIX * VPP + 255) / 256;
    static constexpr int NV_A = (NHALO * VPP + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)NPART * (NPIX + NHALO) * PITCH * 2;
};

template <typename T, int TH, int TW, int NT, int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    using C = WgCfg<T, TH, TW, NT, MODE>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* ld = reinterpret_cast<bf16*>(smem_raw);            // dY tile  [NPIX][PITCH]   (fp32 mode: high halves)
    bf16* la = ld + C::NPIX * C::PITCH;                      // A halo   [NHALO][PITCH]
    bf16* ld_lo = la + C::NHALO * C::PITCH;                  // fp32 mode: the low halves of both tiles
    bf16* la_lo = ld_lo + C::NPIX * C::PITCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ct = blockIdx.x;
    const int co0 = (ct / p.cin_tiles) * 64, ci0 = (ct % p.cin_tiles) * 64;
    const int split = blockIdx.y;
    const int per = (p.ksteps + p.nsplit - 1) / p.nsplit;
    const int s_begin = split * per, s_end = min(p.ksteps, s_begin + per);
    const int tpi = p.tiles_x * p.tiles_y;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

    const bool do_bias = p.bias_ws != nullptr && ci0 == 0 && wn == 0;      // wave-uniform
    f32x16 accb;
#pragma unroll
    for (int j = 0; j < 16; ++j) accb[j] = 0.f;

    // per-image sums: a split is a run of pixel tiles in image order, so its tile of sums is written out and restarted whenever the
    // image changes; the images a split never sees get zeros (every entry of the workspace is written exactly once)
    const int nimg = p.bias_images;
    auto bias_flush = [&](int b) __attribute__((always_inline)) {
        if ((lane & 31) == 0) {
            float* dst = p.bias_ws + ((size_t)(nimg ? split * nimg + b : split)) * p.Cout_pad + co0 + 32 * wm;
#pragma unroll
            for (int j = 0; j < 16; ++j) dst[(j & 3) + 8 * (j >> 2) + 4 * (lane >> 5)] = accb[j];
        }
    };
    int cur_b = s_begin / tpi;
    const int first_b = cur_b;

    u32x4 rd[C::NV_D], ra[C::NV_A];
    auto issue = [&](int step) __attribute__((always_inline)) {
        const int b = step / tpi, r = step - b * tpi;
        const int ty0 = (r / p.tiles_x) * TH, tx0 = (r % p.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < C::NV_D; ++i) {
            const int v = tid + i * 256;
            const int pix = v / C::VPP, cv = v % C::VPP;
            const int y = ty0 + pix / TW, x = tx0 + pix % TW, c = co0 + cv * C::EPV;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (pix < C::NPIX && y < p.Hout && x < p.Wout && c < p.Cout)
                val = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.dy) + ((size_t)(b * p.Hout + y) * p.Wout + x) * p.Cout + c);
            rd[i] = val;
        }
#pragma unroll
        for (int i = 0; i < C::NV_A; ++i) {
            const int v = tid + i * 256;
            const int pix = v / C::VPP, cv = v % C::VPP;
            const int hy = pix / C::HWD, hx = pix % C::HWD;
            int gy, gx;
            if (NT == 1) { gy = ty0 + hy; gx = tx0 + hx; }
            else if (MODE == 2) { gy = 2 * ty0 - 1 + hy; gx = 2 * tx0 - 1 + hx; }
            else { gy = ty0 - 1 + hy; gx = tx0 - 1 + hx; }
            bool ok = pix < C::NHALO && gy >= 0 && gx >= 0;
            if (MODE == 1) { ok = ok && gy < 2 * p.Hin && gx < 2 * p.Win; gy >>= 1; gx >>= 1; }
            else ok = ok && gy < p.Hin && gx < p.Win;
            const int c = ci0 + cv * C::EPV;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (ok && c < p.Cin) {
                const size_t px = (size_t)(b * p.Hin + gy) * p.Win + gx;
                const T* src = c < p.C0 ? reinterpret_cast<const T*>(p.a0) + px * p.C0 + c
                                        : reinterpret_cast<const T*>(p.a1) + px * p.C1 + (c - p.C0);
                val = *reinterpret_cast<const u32x4*>(src);
            }
            ra[i] = val;
        }
    };
    // fp32 mode: four fp32 values -> four bf16 high halves + four bf16 low halves (x - bf16(x)), 8 bytes each
    auto split_store = [&](bf16* hi, bf16* lo, int off, const u32x4& raw) __attribute__((always_inline)) {
        bf16x4 h4, l4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = __uint_as_float(raw[k]);
            h4[k] = (bf16)x;
            l4[k] = (bf16)(x - (float)h4[k]);
        }
        *reinterpret_cast<bf16x4*>(hi + off) = h4;
        *reinterpret_cast<bf16x4*>(lo + off) = l4;
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::NV_D; ++i) {
            const int v = tid + i * 256;
            if (v < C::NPIX * C::VPP) {
                const int off = (v / C::VPP) * C::PITCH + (v % C::VPP) * C::EPV;
                if constexpr (C::F32) split_store(ld, ld_lo, off, rd[i]);
                else *reinterpret_cast<u32x4*>(ld + off) = rd[i];
            }
        }
#pragma unroll
        for (int i = 0; i < C::NV_A; ++i) {
            const int v = tid + i * 256;
            if (v < C::NHALO * C::VPP) {
                const int off = (v / C::VPP) * C::PITCH + (v % C::VPP) * C::EPV;
                if constexpr (C::F32) split_store(la, la_lo, off, ra[i]);
                else *reinterpret_cast<u32x4*>(la + off) = ra[i];
            }
        }
    };

    if (s_begin < s_end) issue(s_begin);
    for (int step = s_begin; step < s_end; ++step) {
        __syncthreads();                                     // the previous step's readers are done with the tiles
        commit();
        __syncthreads();
        if (step + 1 < s_end) issue(step + 1);
        if (do_bias && nimg) {
            const int b_step = step / tpi;
            if (b_step != cur_b) {                           // workgroup-uniform
                bias_flush(cur_b);
#pragma unroll
                for (int j = 0; j < 16; ++j) accb[j] = 0.f;
                cur_b = b_step;
            }
        }
        {
            // k index of an MFMA k-group (16 pixels): k = 8h + j  <->  pixel (row kg, x = 8h + j) for TW = 16,
            //                                                        pixel (row 2kg + h, x = j) for TW = 8
            const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, pq = lane & 3;
            const int py = TW == 16 ? 0 : h, px = (TW == 16 ? 8 * h : 0) + q;      // this lane's address duty, relative to the k-group
            const int coff = 16 * g1 + 4 * pq;
            const int aoff = (py * TW + px) * C::PITCH + 32 * wm + coff;
            const int sp = MODE == 2 ? 2 : 1;
            const int boff = (sp * py * C::HWD + sp * px) * C::PITCH + 32 * wn + coff;
            constexpr int NKG = C::NPIX / 16, RPG = TW == 16 ? 1 : 2;              // k-groups per tile, tile rows per k-group
            typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
            auto frag = [&](const bf16* p0, int second) __attribute__((always_inline)) -> bf16x8 {
                const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + second));
                const s16x8 f = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                return __builtin_bit_cast(bf16x8, f);
            };
            const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};      // bf16 1.0
            const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
#pragma unroll
            for (int kg = 0; kg < NKG; ++kg) {
                const int ao = aoff + kg * RPG * TW * C::PITCH;
                const bf16x8 af = frag(ld + ao, 4 * C::PITCH);
                bf16x8 af_lo = af;
                if constexpr (C::F32) af_lo = frag(ld_lo + ao, 4 * C::PITCH);
                if (do_bias) {
                    if constexpr (C::F32) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_lo, ones, accb, 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ones, accb, 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int ky = NT == 1 ? 0 : t / 3, kx = NT == 1 ? 0 : t % 3;
                    const int bo = boff + ((sp * kg * RPG + ky) * C::HWD + kx) * C::PITCH;
                    const bf16x8 bf = frag(la + bo, 4 * sp * C::PITCH);
                    if constexpr (C::F32) {                   // small terms first
                        const bf16x8 bf_lo = frag(la_lo + bo, 4 * sp * C::PITCH);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_lo, bf, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf_lo, acc[t], 0, 0, 0);
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // partial sums: ws[split][tap][co][ci]; register j of a lane = row (j&3) + 8(j>>2) + 4(lane>>5), column lane & 31
    const int n = ci0 + 32 * wn + (lane & 31);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* dst = p.ws + ((size_t)(split * NT + t) * p.Cout_pad) * p.Cin_pad;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = co0 + 32 * wm + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
            dst[(size_t)m * p.Cin_pad + n] = acc[t][j];
        }
    }
    if (do_bias) {                                           // every column of the ones-tile holds the same sums: column 0 leaves
        if (s_begin < s_end || !nimg) bias_flush(nimg ? cur_b : 0);
        if (nimg) {
#pragma unroll
            for (int j = 0; j < 16; ++j) accb[j] = 0.f;
            for (int b = 0; b < nimg; ++b)
                if (s_begin >= s_end || b < first_b || b > cur_b) bias_flush(b);
        }
    }
}

// dw[co][ci][tap] = sum over splits (in a fixed order) of ws[split][tap][co][ci].
// Two shapes of workgroup, by the number of splits (reduce_ppb):
//   many splits (> 8: the 64 ... 128-channel layers, up to 128 splits of a few hundred KB): 64 consecutive (co, ci) pairs, thread
//     (pair, q) sums the splits s = q, q + 4, ... of every tap, the four partial sums meet in LDS in the order q = 0..3;
//   few splits (<= 8: the wide layers, 2 ... 8 splits of 9 ... 19 MB): 256 consecutive pairs, one thread per pair walks the splits
//     in order with two splits' taps in flight - 1 KiB contiguous per (split, tap) row instead of 256 B, a quarter of the workgroups.
// Either way consecutive threads = consecutive ci (coalesced reads) and the results leave through LDS as one contiguous run of
// the PyTorch layout (coalesced writes).
__host__ __device__ inline int reduce_ppb(int nsplit) { return nsplit <= 8 ? 256 : 64; }

// layout 0: dw [cout][cin][tap] (PyTorch's contiguous weight), element (pair, t) at pair * NT + t; layout 1: dw [cout][tap][cin] (the
// same tensor in channels-last memory order, the form the training step keeps its 3x3 weights in: the re-pack then gathers
// contiguous runs), element at (co * NT + t) * Cin_w + ci.
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ ws, int nsplit, int NT, int Cout_pad, int Cin_pad,
                                                   int Cout_w, int Cin_w, float* __restrict__ dw, int blk, float* sm, int layout) {
    const int64_t npairs = (int64_t)Cout_w * Cin_w;
    const size_t plane = (size_t)Cout_pad * Cin_pad;
    const int ppb = reduce_ppb(nsplit);
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    if (ppb == 256) {
        const int64_t i = (int64_t)blk * 256 + threadIdx.x;
        if (i < npairs) {
            const int co = (int)(i / Cin_w), ci = (int)(i % Cin_w);
            const float* src = ws + (size_t)co * Cin_pad + ci;
            int s = 0;
            for (; s + 1 < nsplit; s += 2) {
                const float* p0 = src + (size_t)s * NT * plane;
                const float* p1 = p0 + (size_t)NT * plane;
                float v0[9], v1[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) { v0[t] = t < NT ? p0[(size_t)t * plane] : 0.f; v1[t] = t < NT ? p1[(size_t)t * plane] : 0.f; }
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = (acc[t] + v0[t]) + v1[t];
            }
            if (s < nsplit) {
                const float* p0 = src + (size_t)s * NT * plane;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (t < NT) acc[t] += p0[(size_t)t * plane];
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) sm[threadIdx.x * 9 + t] = acc[t];
        __syncthreads();
        const int64_t base = (int64_t)blk * 256 * NT, total = npairs * NT;
        if (layout == 0) {
            for (int e = threadIdx.x; e < 256 * NT; e += 256) {
                if (base + e < total) {
                    const int pr = e / NT, t = e - pr * NT;
                    dw[base + e] = sm[pr * 9 + t];
                }
            }
        } else {
            for (int e = threadIdx.x; e < 256 * NT; e += 256) {      // e = t * 256 + pair: consecutive threads, consecutive ci
                const int t = e >> 8, pr = e & 255;
                const int64_t ip = (int64_t)blk * 256 + pr;
                if (ip < npairs) {
                    const int co = (int)(ip / Cin_w), ci = (int)(ip % Cin_w);
                    dw[((int64_t)co * NT + t) * Cin_w + ci] = sm[pr * 9 + t];
                }
            }
        }
        return;
    }
    const int pair = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (int64_t)blk * 64 + pair;
    if (i < npairs) {
        const int co = (int)(i / Cin_w), ci = (int)(i % Cin_w);
        const float* src = ws + (size_t)co * Cin_pad + ci;
        for (int s = q; s < nsplit; s += 4) {
            const float* ps = src + (size_t)s * NT * plane;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if (t < NT) acc[t] += ps[(size_t)t * plane];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) sm[(q * 64 + pair) * 10 + t] = acc[t];
    __syncthreads();
    const int64_t base = (int64_t)blk * 64 * NT, total = npairs * NT;
    for (int e = threadIdx.x; e < 64 * NT; e += 256) {
        int pr, t;
        int64_t idx;
        if (layout == 0) { pr = e / NT; t = e - pr * NT; idx = base + e; }
        else {
            t = e >> 6; pr = e & 63;
            const int64_t ip = (int64_t)blk * 64 + pr;
            idx = ip < npairs ? ((ip / Cin_w) * NT + t) * Cin_w + ip % Cin_w : -1;
        }
        if (layout == 0 ? base + e < total : idx >= 0)
            dw[idx] = ((sm[pr * 10 + t] + sm[(64 + pr) * 10 + t]) + sm[(128 + pr) * 10 + t]) + sm[(192 + pr) * 10 + t];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int nsplit, int NT, int Cout_pad, int Cin_pad,
                                                           int Cout_w, int Cin_w, float* __restrict__ dw, int layout) {
    __shared__ float sm[4 * 64 * 10];
    wgrad_reduce_block(ws, nsplit, NT, Cout_pad, Cin_pad, Cout_w, Cin_w, dw, blockIdx.x, sm, layout);
}

// every layer's reduction in ONE launch (the training step on one GPU defers them to the end of the backward pass: 94 launches
// of 10-20 us each are latency, this one runs at bandwidth).  items[] is sorted by block0 (an item has
// ceil(Cout_w * Cin_w / reduce_ppb(nsplit)) blocks); a block finds its item by bisection on a copy of the block0 column in LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_all_kernel(const hsidm_wgrad_item* __restrict__ items, int n_items) {
    __shared__ float sm[4 * 64 * 10];
    __shared__ int blk0[256];
    int lo = 0, hi = n_items - 1;
    if (n_items <= 256) {
        if ((int)threadIdx.x < n_items) blk0[threadIdx.x] = items[threadIdx.x].block0;
        __syncthreads();
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
    } else {
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (items[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
    }
    const hsidm_wgrad_item it = items[lo];
    wgrad_reduce_block(it.ws, it.nsplit, it.NT, it.Cout_pad, it.Cin_pad, it.Cout_w, it.Cin_w, it.dw, (int)blockIdx.x - it.block0, sm, it.layout);
}

struct WgPlan { int TW, TH, tiles_x, tiles_y, ksteps, nsplit, cin_tiles, cout_tiles, Cin_pad, Cout_pad, NT, mode; size_t ws_bytes, bias_off; };

static int wgrad_plan(int C0, int C1, int B, int Hin, int Win, int Hout, int Wout, int Cout, int ksize, int stride, int ups, WgPlan& pl) {
    const int Cin = C0 + C1;
    if (C0 <= 0 || (C0 & 7) || C1 < 0 || (C1 & 7) || Cout <= 0 || (Cout & 7) || B <= 0 || Hin <= 0 || Win <= 0) return HSIDM_E_BADARG;
    if (ksize != 3 && ksize != 1) return HSIDM_E_BADARG;
    if (stride != 1 && stride != 2) return HSIDM_E_BADARG;
    if (ksize == 1 && (stride != 1 || ups)) return HSIDM_E_UNSUPPORTED;
    if (ups && stride != 1) return HSIDM_E_UNSUPPORTED;
    const int eh = ups ? 2 * Hin : (stride == 2 ? (Hin + 1) / 2 : Hin), ew = ups ? 2 * Win : (stride == 2 ? (Win + 1) / 2 : Win);
    if (eh != Hout || ew != Wout) return HSIDM_E_BADARG;
    pl.NT = ksize * ksize;
    pl.mode = ups ? 1 : (stride == 2 ? 2 : 0);
    pl.TW = Wout >= 16 ? 16 : 8;
    pl.TH = stride == 2 ? 4 : 8;
    pl.tiles_x = (Wout + pl.TW - 1) / pl.TW;
    pl.tiles_y = (Hout + pl.TH - 1) / pl.TH;
    pl.ksteps = B * pl.tiles_x * pl.tiles_y;
    pl.cin_tiles = (Cin + 63) / 64;
    pl.cout_tiles = (Cout + 63) / 64;
    pl.Cin_pad = pl.cin_tiles * 64;
    pl.Cout_pad = pl.cout_tiles * 64;
    const int tiles = pl.cin_tiles * pl.cout_tiles;
    int ns = (2 * device_cus() + tiles - 1) / tiles;                 // about two workgroups per CU
    const size_t per_split = (size_t)pl.NT * pl.Cout_pad * pl.Cin_pad * sizeof(float);
    const size_t cap = (size_t)64 << 20;                              // partial sums are written and read once each
    if ((size_t)ns * per_split > cap) ns = (int)(cap / per_split);
    if (ns > pl.ksteps / 4) ns = pl.ksteps / 4;                      // at least four pixel tiles per workgroup: the partial tile it
    if (ns < 1) ns = 1;                                               // writes (147 KB with 9 taps) must not outweigh what it computes
    const int per = (pl.ksteps + ns - 1) / ns;
    pl.nsplit = (pl.ksteps + per - 1) / per;                          // no empty split
    pl.bias_off = (size_t)pl.nsplit * per_split;                      // [nsplit][Cout_pad] bias partials behind the weight partials
    pl.ws_bytes = pl.bias_off + (size_t)pl.nsplit * B * pl.Cout_pad * sizeof(float);      // room for the per-image form
    return HSIDM_OK;
}

template <typename T, int TH, int TW, int NT, int MODE>
static int launch_wgrad(const WgradParams& p, const WgPlan& pl, hipStream_t s) {
    using C = WgCfg<T, TH, TW, NT, MODE>;
    static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
    static PerDeviceOnce once;
    if (int rc = raise_lds_cap(once, &conv_wgrad_kernel<T, TH, TW, NT, MODE>, C::LDS_BYTES)) return rc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_kernel<T, TH, TW, NT, MODE>), dim3(pl.cin_tiles * pl.cout_tiles, pl.nsplit), dim3(256),
                       C::LDS_BYTES, s, p);
    return (int)hipGetLastError();
}

template <typename T>
static int dispatch_wgrad(const WgradParams& p, const WgPlan& pl, hipStream_t s) {
    const bool w16 = pl.TW == 16;
    if (pl.NT == 1) return w16 ? launch_wgrad<T, 8, 16, 1, 0>(p, pl, s) : launch_wgrad<T, 8, 8, 1, 0>(p, pl, s);
    if (pl.mode == 0) return w16 ? launch_wgrad<T, 8, 16, 9, 0>(p, pl, s) : launch_wgrad<T, 8, 8, 9, 0>(p, pl, s);
    if (pl.mode == 1) return w16 ? launch_wgrad<T, 8, 16, 9, 1>(p, pl, s) : launch_wgrad<T, 8, 8, 9, 1>(p, pl, s);
    return w16 ? launch_wgrad<T, 4, 16, 9, 2>(p, pl, s) : launch_wgrad<T, 4, 8, 9, 2>(p, pl, s);
}

}  // namespace hsidm

using namespace hsidm;

extern "C" int64_t hsidm_conv_wgrad_workspace_bytes(int C0, int C1, int B, int Hin, int Win, int Hout, int Wout, int Cout, int ksize,
                                                    int stride, int ups) {
    WgPlan pl;
    const int rc = wgrad_plan(C0, C1, B, Hin, Win, Hout, Wout, Cout, ksize, stride, ups, pl);
    return rc != HSIDM_OK ? (int64_t)rc : (int64_t)pl.ws_bytes;
}

extern "C" int hsidm_conv_wgrad_plan(int C0, int C1, int B, int Hin, int Win, int Hout, int Wout, int Cout, int ksize, int stride, int ups,
                                     int32_t* plan5) {
    WgPlan pl;
    const int rc = wgrad_plan(C0, C1, B, Hin, Win, Hout, Wout, Cout, ksize, stride, ups, pl);
    if (rc != HSIDM_OK) return rc;
    if (!plan5) return HSIDM_E_BADARG;
    plan5[0] = pl.nsplit; plan5[1] = pl.NT; plan5[2] = pl.Cout_pad; plan5[3] = pl.Cin_pad; plan5[4] = reduce_ppb(pl.nsplit);
    return HSIDM_OK;
}

extern "C" int hsidm_wgrad_reduce_all(const hsidm_wgrad_item* items_dev, int n_items, int total_blocks, void* stream) {
    if (!items_dev || n_items <= 0 || total_blocks <= 0) return HSIDM_E_BADARG;
    hipLaunchKernelGGL(wgrad_reduce_all_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
    return (int)hipGetLastError();
}

extern "C" int hsidm_conv_wgrad(int prec, const void* a0, const void* a1, int C0, int C1, const void* dy, int B, int Hin, int Win,
                                int Hout, int Wout, int Cout, int ksize, int stride, int ups, int Cout_w, int Cin_w, float* dw,
                                int dw_layout, int with_bias, float* db, void* workspace, int64_t workspace_bytes, void* stream) {
    WgPlan pl;
    const int rc = wgrad_plan(C0, C1, B, Hin, Win, Hout, Wout, Cout, ksize, stride, ups, pl);
    if (rc != HSIDM_OK) return rc;
    if (!a0 || (C1 > 0 && !a1) || !dy || !workspace || Cout_w <= 0 || Cout_w > Cout || Cin_w <= 0 || Cin_w > C0 + C1) return HSIDM_E_BADARG;
    if ((size_t)workspace_bytes < pl.ws_bytes) return HSIDM_E_BADARG;
    if (with_bias < 0 || with_bias > 2 || (with_bias && dw && !db) || dw_layout < 0 || dw_layout > 1) return HSIDM_E_BADARG;
    WgradParams p;
    p.a0 = a0; p.a1 = C1 > 0 ? a1 : nullptr; p.dy = dy; p.ws = (float*)workspace;
    p.bias_ws = with_bias ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pl.bias_off) : nullptr;
    p.bias_images = with_bias == 2 ? B : 0;
    p.C0 = C0; p.C1 = C1; p.Cin = C0 + C1; p.Cout = Cout;
    p.B = B; p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout;
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ksteps = pl.ksteps; p.nsplit = pl.nsplit;
    p.cin_tiles = pl.cin_tiles; p.Cin_pad = pl.Cin_pad; p.Cout_pad = pl.Cout_pad;
    hipStream_t s = (hipStream_t)stream;
    int e;
    if (prec == HSIDM_BF16) e = dispatch_wgrad<bf16>(p, pl, s);
    else if (prec == HSIDM_F32X3) e = dispatch_wgrad<float>(p, pl, s);
    else return HSIDM_E_BADARG;
    if (e) return e;
    if (!dw) return HSIDM_OK;                       // deferred: the caller sums the partial tiles later (hsidm_wgrad_reduce_all)
    const int64_t n = (int64_t)Cout_w * Cin_w;
    const int ppb = reduce_ppb(pl.nsplit);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + ppb - 1) / ppb)), dim3(256), 0, s, (const float*)workspace, pl.nsplit, pl.NT,
                       pl.Cout_pad, pl.Cin_pad, Cout_w, Cin_w, dw, dw_layout);
    if (with_bias == 1)                             // the bias partials are a [nsplit][1 tap][Cout_pad][1] stack of the same kind
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((Cout_w + ppb - 1) / ppb)), dim3(256), 0, s, (const float*)p.bias_ws, pl.nsplit,
                           1, pl.Cout_pad, 1, Cout_w, 1, db, 0);
    if (with_bias == 2)                             // per image: [nsplit][1][B * Cout_pad][1] -> db [B][Cout_pad]
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((B * pl.Cout_pad + ppb - 1) / ppb)), dim3(256), 0, s, (const float*)p.bias_ws,
                           pl.nsplit, 1, B * pl.Cout_pad, 1, B * pl.Cout_pad, 1, db, 0);
    return (int)hipGetLastError();
}
