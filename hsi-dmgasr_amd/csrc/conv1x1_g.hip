// conv1x1_g: LDS-staged 1x1 convolution (a GEMM out[pixel][cout] = T(x[pixel][:]) . W[cout][:]) for the bf16
// throughput mode: ResnetBlock residual projections (reference unet.py:102-103), attention qkv / out projections
// (unet.py:128,141).
//
// Measurements behind it (tools/conv_bench.py, batch 120): the generic v1 kernel ran these GEMMs at 180-310 TFLOP/s,
// the weight-stationary kernel (conv1x1_ws.hip) loads activations as MFMA A-fragments (16 B per lane, 32 different
// rows per wave-load) and is bound by the texture addresser, 1.6x off the HBM floor on the 128x128 level.  Here
//   * the activation tile (128 consecutive pixels x 64 channels) is fetched with row-contiguous 16-byte loads (8 lanes
//     = one 128-byte line), held in registers for one K chunk and a half, and committed to a double-buffered padded
//     LDS tile (160-byte pixel rows, the conv_v2 layout: conflict-free ds_read_b128 A operands of v_mfma_f32_16x16x32, the matrix
//     instruction of the conv kernels since round 3 - conv_v2.h; weights are 16-byte units of the same packed order);
//   * weights never touch LDS: each wave streams its B-fragments from L2 through an 8-slot register ring, six
//     fragments (1.5 K chunks) ahead, in the order conv_v2 uses (one wave-load = 1 KiB contiguous);
//   * the workgroup is persistent over (pixel tile, cout slice) items with conv_v2's XCD-aware slice mapping; the
//     staging of the next item's first chunks overlaps the epilogue;
//   * optional GroupNorm affine on the input (attention qkv: norm without SiLU) in the commit step;
//   * epilogue as in conv_v2: bias, LDS transposition, 16-byte stores, residual, statistics per 64-pixel group.
// K is padded to a multiple of 128 with zero weights (two chunks per loop trip keep every register index static).
//
// IM (im2col-8): the same GEMM for a 3x3 stride-1 convolution whose input has exactly 8 channels (the UNet stem conv,
// reference unet.py:177-178: 6 -> 64, padded to 8).  On conv_v2 those 8 channels occupy a 64-channel chunk (36 k-slices of
// which 31.5 multiply zeros: 234 us at batch 120 for a layer whose HBM floor is 45 us).  Here the K axis is tap-major,
// k = 8*tap + c: staged vector (pixel, tap) is the 16-byte pixel in[y+dy][x+dx][0..7] (zero outside the image), so the
// layer is a K = 72 -> 128 GEMM: 2 chunks, 8 k-slices.
#include "conv_v2.h"
#include "../../include/hsidm.h"
#include <type_traits>

// Diagnostic builds (-DG1_ABL=n, tools/g1_ablate.sh): 1 no activation loads, 2 no weight loads, 3 no MFMAs, 4 no A-fragment LDS
// reads, 5 no barriers, 6 no output stores, 7 no epilogue, 8 no statistics (results are wrong; timing only).  The product build has
// G1_ABL = 0 and none of this code.  What it showed at batch 240 (profiles/r02_small_batch/g1_ablate.txt): no single piece of the K
// loop is worth more than 5 % (the MFMAs: 93 -> 89 us on the 1024 -> 512 projection) except the activation loads (-21 %), and the
// epilogue as a whole is 20-40 % of a launch (qkv 128 -> 76 us without it; its stores alone 7-14 %): per item ~6.5 us against
// 1.15 us per 64-channel chunk.  Tried on that evidence and dropped: the tile transposed by swapping the MFMA operands, stored
// straight from the accumulators (8 bytes per lane: slower, 167 vs 147 us on qkv) or through a patch written with two
// ds_write_b128 per lane and tile instead of sixteen ds_write_b16 (no change: the b16 writes are not what the epilogue costs).
#ifndef G1_ABL
#define G1_ABL 0
#endif

namespace hsidm {

struct C1gParams {
    const bf16* src0;
    const bf16* src1;
    int C0, C1;
    const f32x4* gn_ab;     // [B][Ctot/2] = (scale, shift) of two channels, or null
    const bf16* w;          // [chunk][Cout_pad/32][kk 4][lane 64][8]
    const bf16* w_lo;       // NP = 2: low halves of the weights, same layout (conv_v2.h, V2Cfg); else null
    const float* bias;
    const float* bias1;     // PAIR kernels: the FIRST convolution's bias (`bias` is the second's)
    int pair_act;           // PAIR kernels: activation between the two convolutions (ACT_NONE | ACT_LEAKY)
    const bf16* res;
    float res_scale;
    bf16* out;
    float2* stats;          // [B][HW/64][Cout] or null
    int M, HW, Cout, Cout_pad, nch;     // nch = K_pad / 64, even
    int im_H, im_W;                     // IM kernels: image size (HW = im_H * im_W)
    int n_slices, m_tiles, total_items;
};

// F32 (with E = bf16, NP = 2): the fp32 mode on this kernel - fp32 activations in HBM (two 16-byte vectors per staged pixel-vector, fp32
// GroupNorm pairs, fp32 stores), every staged value split into bf16 hi + lo into TWO tiles per buffer, three MFMAs per product
// (conv_v2.h, AP = 2); the 4-byte epilogue patches (40 KB) get a region of their own: 112 KB of LDS, one workgroup per CU.
// F32 with E = f16, NP = 2: the "fp32h" kernel set (precision.py) - fp32 storage, fp32 GroupNorm pairs and stores, but ONE fp16 activation
// operand (the staged value rounded once) against fp16 hi + lo weights: two MFMAs per product, one tile per buffer, 80 KB of LDS - two
// workgroups per CU again.
// PAIR (BN = 64, Cin = Cout = 64, XF_NONE): TWO 1x1 convolutions in one launch, out = W2 act(W1 x + b1) + b2 - the body of the group
// autoencoder's spectral ResAttentionBlock (common.py:250-271 with kernel_size 1: conv, LeakyReLU, conv; AE.py:102-109).  The item's two
// 64-channel "chunks" are the two convolutions: chunk 0 multiplies the staged x tile by W1 (weight step 0); its accumulators - plus b1,
// through the activation, converted to the operand type (fp32 form: split into bf16 hi + lo, exactly what staging a stored fp32 h would
// have produced) - are written straight into LDS buffer 1 as the A operand of chunk 1, which multiplies by W2 (weight step 1).  h never
// exists in HBM (2 of the 7 GB a block moved), one launch instead of two, and neither GEMM pays the zero chunk that pads K = 64 to the
// two-chunk trip of the plain kernel (half of its matrix instructions on these layers).  Epilogue (b2, statistics for the CALayer's
// global average, store) as before.
template <int BN, int XF, int IM = 0, typename E = bf16, int NP = 1, bool F32 = false, bool PAIR = false>
__global__ __launch_bounds__(256, (F32 && __is_same(E, bf16)) ? 1 : 2) void conv1x1_g_kernel(const C1gParams p) {
    static_assert(!PAIR || (BN == 64 && XF == XF_NONE && IM == 0), "the 1x1 pair: 64 -> 64 -> 64, no input transform");
    using EL = Elem<E>;
    using x8 = typename EL::x8;
    using x2 = typename EL::x2;
    using S = typename std::conditional<F32, float, E>::type;      // storage type of activations in HBM
    static_assert(!F32 || (NP == 2 && IM == 0), "the fp32 form: hi + lo weights, plain 1x1");
    constexpr int SV = F32 ? 2 : 1, AP = (F32 && __is_same(E, bf16)) ? 2 : 1;      // (activation operands per staged value: hi [+ lo])
    constexpr int WN = BN / 32, WM = 4 / WN, MR = 128 / WM / 32;
    constexpr int PSTR = 80, TILE = 128 * PSTR, BUFE = AP * TILE;
    constexpr int SCR_STR = 40;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* xt = reinterpret_cast<E*>(smem_raw);                     // [2][AP][TILE] (+ 2 KiB: the 16-bit epilogue patch overruns buffer 1)
    if constexpr (!__is_same(E, bf16)) fp16_saturating_stores();        // (common.h)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lc = lane & 15, lg = lane >> 4;                   // MFMA 16x16x32: row / column lc, k group lg (conv_v2.h)
    const int G = gridDim.x;
    const int ctot = p.C0 + p.C1;

    int item = blockIdx.x;                                      // item = m_tile * n_slices + n_slice
    const int ns = item % p.n_slices;                           // constant for this block (G % n_slices == 0)
    const int n0 = ns * BN;
    const int n_items_blk = (p.total_items - item + G - 1) / G;

    // ---- weight stream -------------------------------------------------------------------------------------------
    const int nsw = p.Cout_pad >> 5;
    const size_t wl_off = ((size_t)(ns * WN + wn) * 4 * 64 + (lg >> 1) * 64 + 32 * (lg & 1) + lc) * 8;     // (conv_v2.h: wlane_off, frag_off)
    const E* wlane = reinterpret_cast<const E*>(p.w) + wl_off;
    const E* wlane_lo = reinterpret_cast<const E*>(NP == 2 ? p.w_lo : p.w) + wl_off;
    auto frag_off = [](int e) __attribute__((always_inline)) -> int { return (e >> 1) * (2 * 64 * 8) + (e & 1) * (16 * 8); };
    const size_t wstep_stride = (size_t)nsw * 4 * 64 * 8;
    x8 fring[8], fring_lo[NP == 2 ? 8 : 1];
    int wnext = 0;
    auto f_issue = [&](int slot, int kk) __attribute__((always_inline)) {
        if (G1_ABL != 2) {
            fring[slot] = *reinterpret_cast<const x8*>(wlane + (size_t)wnext * wstep_stride + frag_off(kk));
            if (NP == 2) fring_lo[NP == 2 ? slot : 0] = *reinterpret_cast<const x8*>(wlane_lo + (size_t)wnext * wstep_stride + frag_off(kk));
        }
        if (kk == 3) wnext = (wnext + 1 == p.nch) ? 0 : wnext + 1;
    };

    // ---- activation staging: vector i of this thread = pixel i*32 + tid/8 of the tile, channels 8*(tid%8).. of the chunk
    const int cv = tid & 7;
    const int px_l = tid >> 3;
    const int hv0 = px_l * PSTR + cv * 8;
    u32x4 hreg[2][4][SV];
    int set_m0[2] = {0, 0}, set_cc[2] = {0, 0};
    unsigned set_zero[2] = {0u, 0u};                           // IM: bit i = staged vector i lies outside the image
    bool set_ok[2] = {false, false};
    int st_item = item, st_chunk = 0;
    bool st_valid = true;
    // first pixel of an item's tile: item / n_slices is a ~20-instruction scalar division, and it was paid per staged chunk;
    // the block's items are G apart and G % n_slices == 0, so the tile advances by a constant
    const int m0_step = (G / p.n_slices) * 128;
    int st_m0 = (item / p.n_slices) * 128, cur_m0 = st_m0;
    auto issue = [&](int S_) __attribute__((always_inline)) {
        if (PAIR && S_ == 1) {                                  // an item's chunk 1 is made in the kernel (h): nothing to request
            set_ok[1] = false;
            if (st_valid) { st_chunk = 0; st_item += G; st_m0 += m0_step; st_valid = st_item < p.total_items; }
            return;
        }
        set_ok[S_] = st_valid;
        if (!st_valid) return;
        const int m0 = st_m0;
        const int c = st_chunk * 64 + cv * 8;
        const int cc = c < ctot ? c : 0;                        // zero-weight padding: any finite data will do
        const S* src;
        int cs;
        if (cc < p.C0) { src = reinterpret_cast<const S*>(p.src0) + cc; cs = p.C0; }
        else           { src = reinterpret_cast<const S*>(p.src1) + (cc - p.C0); cs = p.C1; }
        set_m0[S_] = m0;
        set_cc[S_] = cc;
        if (IM) {
            const int tap = st_chunk * 8 + cv;                  // k = 8*tap + channel: this thread's vector is one tap of one pixel
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            unsigned zero = 0u;
            // position of the tile's first pixel: once per item on the scalar unit; im_W is a power of two and HW >= 128,
            // so a vector's (y, x) follows with shifts and at most one image wrap
            const int r0 = m0 % p.HW;
            const int wsh = 31 - __builtin_clz(p.im_W);
            const int y0 = r0 >> wsh, x0 = r0 & (p.im_W - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int m = m0 + i * 32 + px_l;
                m = m < p.M ? m : p.M - 1;
                const int xl = x0 + (m - m0);
                int y = y0 + (xl >> wsh);
                const int x = xl & (p.im_W - 1);
                y = y >= p.im_H ? y - p.im_H : y;
                const int yy = y + dy, xx = x + dx;
                const bool ok = tap < 9 && yy >= 0 && yy < p.im_H && xx >= 0 && xx < p.im_W;
                zero |= ok ? 0u : (1u << i);
                const int ms = ok ? m + dy * p.im_W + dx : m;
                hreg[S_][i][0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const E*>(p.src0) + (size_t)ms * 8);
            }
            set_zero[S_] = zero;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int m = m0 + i * 32 + px_l;
                m = m < p.M ? m : p.M - 1;
                if (G1_ABL != 1) {
                    const u32x4* sv = reinterpret_cast<const u32x4*>(src + (size_t)m * cs);
#pragma unroll
                    for (int h = 0; h < SV; ++h) hreg[S_][i][h] = sv[h];
                }
            }
        }
        if (++st_chunk == p.nch) { st_chunk = 0; st_item += G; st_m0 += m0_step; }
        st_valid = st_item < p.total_items;
    };
    unsigned abh[2][8];                                         // GroupNorm (scale, shift), fp16x2, for pixel halves 0-63 / 64-127
    float gsc[F32 ? 2 : 1][8], gsh[F32 ? 2 : 1][8];             // fp32 form: the fp32 pairs
    auto params_fetch = [&](int S_) __attribute__((always_inline)) {
        if (XF == XF_NONE || !set_ok[S_]) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = set_m0[S_] + 64 * h;
            m = m < p.M ? m : p.M - 1;
            const int b = m / p.HW;
            const int nb = p.M / p.HW;                          // fp16x2 half of the GroupNorm table (conv_v2.h: gn_params)
            if constexpr (F32) {
                const f32x4* tf = p.gn_ab + (((size_t)b * ctot + set_cc[S_]) >> 1);
                f32x4 r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = tf[k];
#pragma unroll
                for (int k = 0; k < 4; ++k) { gsc[h][2 * k] = r[k][0]; gsh[h][2 * k] = r[k][1]; gsc[h][2 * k + 1] = r[k][2]; gsh[h][2 * k + 1] = r[k][3]; }
                continue;
            }
            const u32x4* t = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned*>(p.gn_ab) + (size_t)2 * nb * ctot + (size_t)b * ctot + set_cc[S_]);
            const u32x4 lo = t[0], hi = t[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) { abh[h][k] = lo[k]; abh[h][4 + k] = hi[k]; }
        }
    };
    auto commit = [&](int S_, int i, int buf) __attribute__((always_inline)) {
        if (!set_ok[S_]) return;
        if constexpr (F32) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = __uint_as_float(hreg[S_][i][0][k]); v[4 + k] = __uint_as_float(hreg[S_][i][1][k]); }
            if (XF != XF_NONE) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], gsc[i >> 1][k], gsh[i >> 1][k]);
            }
            x8 o, ol;
#pragma unroll
            for (int k = 0; k < 8; ++k) { o[k] = (E)v[k]; ol[k] = (E)(v[k] - (float)o[k]); }
            *reinterpret_cast<x8*>(xt + buf * BUFE + hv0 + i * 32 * PSTR) = o;
            if constexpr (AP == 2) *reinterpret_cast<x8*>(xt + buf * BUFE + TILE + hv0 + i * 32 * PSTR) = ol;
            return;
        }
        u32x4 raw = hreg[S_][i][0];
        if (IM) {
            const bool z = (set_zero[S_] >> i) & 1u;
#pragma unroll
            for (int k = 0; k < 4; ++k) raw[k] = z ? 0u : raw[k];
        }
        if (XF != XF_NONE) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] = EL::lo(raw[k]);
                v[2 * k + 1] = EL::hi(raw[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], h2_lo(abh[i >> 1][k]), h2_hi(abh[i >> 1][k]));
            x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (E)v[k];       // (GroupNorm affine without an activation: not bounded)
            *reinterpret_cast<x8*>(xt + buf * BUFE + hv0 + i * 32 * PSTR) = o;
        } else {
            *reinterpret_cast<u32x4*>(xt + buf * BUFE + hv0 + i * 32 * PSTR) = raw;
        }
    };

    // ---- MFMA fragment bases ---------------------------------------------------------------------------------------
    int abase[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) abase[mr] = (wm * (128 / WM) + mr * 32 + lc) * PSTR + 8 * lg;     // 16-pixel half r: + 16 r PSTR
    f32x4 acc[MR][2][2];                                        // [32-pixel group][16-pixel half][16-cout half]

    // prologue: chunks 0 and 1 requested, six weight fragments in flight, chunk 0 committed
#pragma unroll
    for (int f = 0; f < 6; ++f) f_issue(f, f % 4);
    issue(0);
    issue(1);
    params_fetch(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) commit(0, i, 0);
    lds_barrier();

    const int n_lane = n0 + wn * 32 + lc;                       // couts n_lane, n_lane + 16 (Cout % BN == 0 on this path)
    float bias[2] = {p.bias ? p.bias[n_lane] : 0.f, p.bias ? p.bias[n_lane + 16] : 0.f};
    float bias1[2] = {(PAIR && p.bias1) ? p.bias1[n_lane] : 0.f, (PAIR && p.bias1) ? p.bias1[n_lane + 16] : 0.f};
    // The bias registers are USED here, once: left pending, their loads kept the compiler's in-order memory counter "unknown" at the loop
    // head, and the first instruction of every epilogue that read them waited s_waitcnt vmcnt(0) - for the weight ring and the next item's
    // staged chunks, i.e. a full round trip exposed per item (found in the ISA; round 2's ablation saw it as "the epilogue is 20-40 % of a launch")
    asm volatile("" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias1[0]), "+v"(bias1[1]));

    for (int it = 0; it < n_items_blk; ++it, item += G) {
        for (int chunk = 0; chunk < p.nch; chunk += 2) {
            // body(PAR): MFMAs on buffer PAR, commits the chunk held in register set PAR^1 into buffer PAR^1, requests the
            // chunk after that into set PAR (whose previous content was committed one body ago)
            auto body = [&](auto par_tag) __attribute__((always_inline)) {
                constexpr int PAR = decltype(par_tag)::value;
                const E* hb = xt + PAR * BUFE;
                params_fetch(PAR ^ 1);                          // before the data requests: a later wait for the parameters then
                issue(PAR);                                     // leaves those (and the weight ring) in flight
                x8 a[3][MR], a_lo[AP == 2 ? 3 : 1][MR];
                auto a_fetch = [&](int u) __attribute__((always_inline)) {          // sub-step u = 2 q + r: 32-channel slice q, pixel half r
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
                        if (G1_ABL != 4) {
                            a[u % 3][mr] = *reinterpret_cast<const x8*>(hb + abase[mr] + (u >> 1) * 32 + (u & 1) * 16 * PSTR);
                            if constexpr (AP == 2) a_lo[u % 3][mr] = *reinterpret_cast<const x8*>(hb + TILE + abase[mr] + (u >> 1) * 32 + (u & 1) * 16 * PSTR);
                        }
                };
                a_fetch(0);
                a_fetch(1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int u = PAR * 4 + kk;                 // position in the two-chunk trip: the weight ring's period
                    const int q = kk >> 1, r = kk & 1;
                    if (kk + 2 < 4) a_fetch(kk + 2);
                    f_issue((u + 6) % 8, (u + 6) % 4);
                    // first use of these accumulators: C = 0 as the MFMA's inline constant, behind a uniform branch (a select between
                    // the constant and the accumulator costs a v_cndmask per register - and, in the fp32 form, moves through the AGPRs)
                    if ((PAR == 0 || PAIR) && kk < 2 && chunk == 0) {     // (PAIR: the second convolution starts from zero too)
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh) acc[mr][r][nh] = EL::mfma16(a[kk % 3][mr], fring[(PAR * 4 + q * 2 + nh) % 8], zero);
                    } else if (G1_ABL != 3) {
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
                                acc[mr][r][nh] = EL::mfma16(a[kk % 3][mr], fring[(PAR * 4 + q * 2 + nh) % 8], acc[mr][r][nh]);
                    }
                    if (NP == 2 && G1_ABL != 3) {
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
                                acc[mr][r][nh] = EL::mfma16(a[kk % 3][mr], fring_lo[NP == 2 ? (PAR * 4 + q * 2 + nh) % 8 : 0], acc[mr][r][nh]);
                    }
                    if constexpr (AP == 2) {                    // third pass: the activations' low halves on the weights' high halves
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
                                acc[mr][r][nh] = EL::mfma16(a_lo[kk % 3][mr], fring[(PAR * 4 + q * 2 + nh) % 8], acc[mr][r][nh]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (kk == 1) { commit(PAR ^ 1, 0, PAR ^ 1); commit(PAR ^ 1, 1, PAR ^ 1); }
                    if (kk == 2) { commit(PAR ^ 1, 2, PAR ^ 1); commit(PAR ^ 1, 3, PAR ^ 1); }
                }
                if (G1_ABL != 5) lds_barrier();
            };
            body(SlotTag<0>{});
            if constexpr (PAIR) {
                // h = act(W1 x + b1) of this wave's 64 pixels x 32 couts -> LDS buffer 1 in the staged layout (pixel rows of PSTR elements):
                // the lane holds pixels 4 lg .. + 3 of couts 16 nh + lc per (32-pixel group, 16-pixel half).  (Buffer 1 carried the previous
                // item's epilogue patch: that item ended with a barrier; nothing was committed to it during body 0.)
                E* hb = xt + BUFE;
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float v = acc[mr][r][nh][j] + bias1[nh];
                                if (p.pair_act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
                                const int at = (wm * (128 / WM) + mr * 32 + 16 * r + 4 * lg + j) * PSTR + wn * 32 + 16 * nh + lc;
                                const E hi = (E)v;
                                hb[at] = hi;
                                if constexpr (AP == 2) hb[TILE + at] = (E)(v - (float)hi);
                            }
                lds_barrier();
            }
            body(SlotTag<1>{});
        }

        // ---- epilogue: buffer 1 is free (the last chunk of an item has odd parity), buffer 0 holds the next item's chunk 0
        const int m0 = cur_m0;
        cur_m0 += m0_step;
        // 16-bit: the patch sits in buffer 1 (free here); fp32 form: 4-byte patches in a region of their own behind the buffers
        S* scr = (F32 ? reinterpret_cast<S*>(xt + 2 * BUFE) : reinterpret_cast<S*>(xt + TILE)) + wave * (64 * SCR_STR);
        int lane_e = lane_id_now();   // rebuilt here, not kept (conv_v2.h)
        asm volatile("" : "+v"(lane_e));
        const int pl0 = lane_e >> 2, cq = lane_e & 3;
        const int lc_e = lane_e & 15, lg_e = lane_e >> 4;
        auto run = [&](auto res_tag) __attribute__((always_inline)) {
            constexpr bool RES = decltype(res_tag)::value != 0;
#pragma unroll
            for (int g = 0; g < MR; g += 2) {
                const int nm = (MR - g) < 2 ? (MR - g) : 2;
                const int mp = m0 + wm * (128 / WM) + g * 32;          // first pixel of this 64-pixel pass (M % 64 == 0)
                if (mp >= p.M) break;
                const size_t obase = (size_t)mp * p.Cout + n0 + wn * 32;
                const unsigned lane_el = (unsigned)(pl0 * p.Cout + cq * 8);
                x8 rv[(RES && !F32) ? 4 : 1];
                f32x4 rvf[(RES && F32) ? 4 : 1][2];
                if (RES) {
#pragma unroll
                    for (int v4 = 0; v4 < 4; ++v4)
                        if (v4 < 2 * nm) {
                            const S* rp = reinterpret_cast<const S*>(p.res) + obase + (size_t)16 * v4 * p.Cout + lane_el;
                            if constexpr (F32) { rvf[RES ? v4 : 0][0] = *reinterpret_cast<const f32x4*>(rp); rvf[RES ? v4 : 0][1] = *reinterpret_cast<const f32x4*>(rp + 4); }
                            else rv[RES ? v4 : 0] = *reinterpret_cast<const x8*>(rp);
                        }
                }
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    if (m2 >= nm) break;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {                    // pixels j, j + 1 of the lane's four: one packed conversion
                                const int at = (m2 * 32 + 16 * r + 4 * lg_e + j) * SCR_STR + 16 * nh + lc_e;
                                if constexpr (F32) {
                                    scr[at] = acc[g + m2][r][nh][j] + bias[nh];
                                    scr[at + SCR_STR] = acc[g + m2][r][nh][j + 1] + bias[nh];
                                } else {
                                    const x2 pr = cvt_pair_hw<E>(acc[g + m2][r][nh][j] + bias[nh], acc[g + m2][r][nh][j + 1] + bias[nh]);
                                    reinterpret_cast<E*>(scr)[at] = pr[0];
                                    reinterpret_cast<E*>(scr)[at + SCR_STR] = pr[1];
                                }
                            }
                }
                float vs1[8], vs2[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) vs1[k] = vs2[k] = 0.f;
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    if (v4 >= 2 * nm) break;
                    float f[8];
                    if constexpr (F32) {
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(scr + (pl0 + 16 * v4) * SCR_STR + cq * 8);
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(scr + (pl0 + 16 * v4) * SCR_STR + cq * 8 + 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { f[k] = r0[k]; f[4 + k] = r1[k]; }
                        if (RES) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) f[k] = fmaf(p.res_scale, f[k], rvf[RES ? v4 : 0][k >> 2][k & 3]);
                        }
                        f32x4 o0, o1;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { o0[k] = f[k]; o1[k] = f[4 + k]; }
                        S* op = reinterpret_cast<S*>(p.out) + obase + (size_t)16 * v4 * p.Cout + lane_el;
                        if (G1_ABL != 6) { *reinterpret_cast<f32x4*>(op) = o0; *reinterpret_cast<f32x4*>(op + 4) = o1; }
                    } else {
                    const x8 raw = *reinterpret_cast<const x8*>(reinterpret_cast<const E*>(scr) + (pl0 + 16 * v4) * SCR_STR + cq * 8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = (float)raw[k];
                    x8 o = raw;
                    if (RES) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            f[k] = fmaf(p.res_scale, f[k], (float)rv[RES ? v4 : 0][k]);   // statistics from the fp32 sum (the store's rounding noise is zero-mean)
                            o[k] = (E)f[k];
                        }
                    }
                    if (G1_ABL != 6) *reinterpret_cast<x8*>(reinterpret_cast<E*>(p.out) + obase + (size_t)16 * v4 * p.Cout + lane_el) = o;
                    }
                    if (p.stats) {          // (uniform: 22 of a step's 28 GEMM launches - projections, qkv - emit no statistics: no unpack, no sums)
#pragma unroll
                        for (int k = 0; k < 8; ++k) { vs1[k] += f[k]; vs2[k] = fmaf(f[k], f[k], vs2[k]); }
                    }
                }
                if (p.stats && G1_ABL != 8) {
                    // halving butterfly over the 16 lanes that hold the same 8 couts (see conv_v2.h)
                    const bool hi0 = (lane_e & 4) != 0, hi1 = (lane_e & 8) != 0, hi2 = (lane_e & 16) != 0, hi3 = (lane_e & 32) != 0;
                    float a8[8], a4[4], a2[2];
#pragma unroll
                    for (int i = 0; i < 8; ++i) a8[i] = (hi0 ? vs2[i] : vs1[i]) + lane_xor<4>(hi0 ? vs1[i] : vs2[i], lane_e);
#pragma unroll
                    for (int i = 0; i < 4; ++i) a4[i] = (hi1 ? a8[i + 4] : a8[i]) + lane_xor<8>(hi1 ? a8[i] : a8[i + 4], lane_e);
#pragma unroll
                    for (int i = 0; i < 2; ++i) a2[i] = (hi2 ? a4[i + 2] : a4[i]) + lane_xor<16>(hi2 ? a4[i] : a4[i + 2], lane_e);
                    const float a1 = (hi3 ? a2[1] : a2[0]) + lane_xor<32>(hi3 ? a2[0] : a2[1], lane_e);
                    const int idx = ((lane_e >> 2) & 1) * 8 + ((lane_e >> 3) & 1) * 4 + ((lane_e >> 4) & 1) * 2 + (lane_e >> 5);
                    const int grp = mp >> 6;                           // 64-pixel group: image = grp / (HW/64)
                    float* dst = reinterpret_cast<float*>(p.stats + (size_t)grp * p.Cout + n0 + wn * 32);
                    dst[(cq * 8 + (idx & 7)) * 2 + (idx >> 3)] = a1;
                }
            }
        };
        if (G1_ABL != 7) { if (p.res) run(SlotTag<1>{}); else run(SlotTag<0>{}); }
        lds_barrier();                                                  // the patch is the next odd chunk's buffer
    }
}


template <int BN, int XF, int IM, typename E, int NP, bool F32 = false, bool PAIR = false>
static int run_g1(C1gParams& p, hipStream_t s) {
    constexpr int AP = (F32 && __is_same(E, bf16)) ? 2 : 1;
    constexpr size_t lds = F32 ? (size_t)2 * AP * 128 * 80 * 2 + (size_t)4 * 64 * 40 * 4 : (size_t)2 * 128 * 80 * 2 + 2048;
    static_assert(AP == 2 || lds <= 80 * 1024, "two workgroups per CU");
    static PerDeviceOnce once;
    if (int rc = raise_lds_cap(once, &conv1x1_g_kernel<BN, XF, IM, E, NP, F32, PAIR>, lds)) return rc;
    const int g1_slots = (AP == 2 ? 1 : 2) * device_cus();
    p.n_slices = p.Cout_pad / BN;
    p.m_tiles = (p.M + 127) / 128;
    p.total_items = p.m_tiles * p.n_slices;
    int lcm = 8;
    while (lcm % p.n_slices) lcm += 8;
    const int slots = g1_slots;
    int G = (p.total_items < slots ? p.total_items : slots) / lcm * lcm;
    if (G == 0) G = p.total_items;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_g_kernel<BN, XF, IM, E, NP, F32, PAIR>), dim3(G), dim3(256), lds, s, p);
    return (int)hipGetLastError();
}

template <typename E>
static int dispatch_g1_f32(int bn, int xf, C1gParams& p, hipStream_t s) {
    if (bn == 128) return xf == XF_NONE ? run_g1<128, XF_NONE, 0, E, 2, true>(p, s) : run_g1<128, XF_AFFINE, 0, E, 2, true>(p, s);
    if (bn == 64) return xf == XF_NONE ? run_g1<64, XF_NONE, 0, E, 2, true>(p, s) : run_g1<64, XF_AFFINE, 0, E, 2, true>(p, s);
    return HSIDM_E_UNSUPPORTED;
}

template <typename E, int NP>
static int dispatch_g1(int bn, int xf, int im, C1gParams& p, hipStream_t s) {
    if (im) {
        if (bn == 128) return run_g1<128, XF_NONE, 1, E, NP>(p, s);
        if (bn == 64) return run_g1<64, XF_NONE, 1, E, NP>(p, s);
        return HSIDM_E_UNSUPPORTED;
    }
    if (bn == 128) return xf == XF_NONE ? run_g1<128, XF_NONE, 0, E, NP>(p, s) : run_g1<128, XF_AFFINE, 0, E, NP>(p, s);
    if (bn == 64) return xf == XF_NONE ? run_g1<64, XF_NONE, 0, E, NP>(p, s) : run_g1<64, XF_AFFINE, 0, E, NP>(p, s);
    return HSIDM_E_UNSUPPORTED;
}

// bn: 64 or 128 (Cout % bn == 0); xf: XF_NONE or XF_AFFINE; HW % 64 == 0; nch even.
// im_W > 0: 3x3 conv of an 8-channel input as a tap-major GEMM (C0 = 8, C1 = 0, xf = XF_NONE, nch = 2).
// elem: 0 bf16, 1 fp16 (the bf16-typed pointers are then fp16 data), 2 the fp32 mode (fp32 tensors, bf16 hi + lo weights: w_lo required),
// 3 the "fp32h" set (fp32 tensors, one fp16 activation operand, fp16 hi + lo weights: w_lo required);
// w_lo != null (fp16): second pass on the weights' low halves
int conv1x1_g_run(int bn, int xf, const bf16* src0, const bf16* src1, int C0, int C1, const void* gn_ab, const bf16* w, const bf16* w_lo,
                  int elem, const float* bias, const bf16* res, float res_scale, bf16* out, float2* stats, int M, int HW, int Cout,
                  int nch, int im_H, int im_W, hipStream_t s) {
    C1gParams p;
    p.im_H = im_H; p.im_W = im_W;
    if (im_W > 0) {
        if (xf != XF_NONE || C0 != 8 || C1 != 0 || nch != 2 || (im_W & (im_W - 1)) || HW < 128) return HSIDM_E_UNSUPPORTED;
    }
    p.src0 = src0; p.src1 = src1; p.C0 = C0; p.C1 = C1;
    p.gn_ab = reinterpret_cast<const f32x4*>(gn_ab);
    p.w = w; p.w_lo = w_lo; p.bias = bias; p.res = res; p.res_scale = res_scale; p.out = out; p.stats = stats;
    p.bias1 = nullptr; p.pair_act = ACT_NONE;
    p.M = M; p.HW = HW; p.Cout = Cout; p.Cout_pad = Cout; p.nch = nch;
    if (elem == 2) return (w_lo && im_W <= 0) ? dispatch_g1_f32<bf16>(bn, xf, p, s) : HSIDM_E_UNSUPPORTED;
    if (elem == 3) return (w_lo && im_W <= 0) ? dispatch_g1_f32<f16>(bn, xf, p, s) : HSIDM_E_UNSUPPORTED;
    if (elem == 0) return w_lo ? HSIDM_E_UNSUPPORTED : dispatch_g1<bf16, 1>(bn, xf, im_W > 0, p, s);
    if (elem == 1) return w_lo ? dispatch_g1<f16, 2>(bn, xf, im_W > 0, p, s) : dispatch_g1<f16, 1>(bn, xf, im_W > 0, p, s);
    return HSIDM_E_UNSUPPORTED;
}

// out = W2 act(W1 x + b1) + b2 on [M][64] tensors (PAIR kernels above): w / w_lo = the two convolutions' weights as the two steps of the
// packed order; elem: 1 fp16 (hi + lo weights: w_lo required), 2 the fp32 mode (fp32 tensors, bf16 hi + lo weights)
int conv1x1_pair_run(const void* x, const bf16* w, const bf16* w_lo, int elem, const float* bias1, int act, const float* bias2, void* out,
                     float2* stats, int M, int HW, hipStream_t s) {
    if (!w_lo || (elem != 1 && elem != 2)) return HSIDM_E_UNSUPPORTED;
    C1gParams p;
    p.im_H = p.im_W = 0;
    p.src0 = reinterpret_cast<const bf16*>(x); p.src1 = nullptr; p.C0 = 64; p.C1 = 0;
    p.gn_ab = nullptr;
    p.w = w; p.w_lo = w_lo; p.bias = bias2; p.bias1 = bias1; p.pair_act = act;
    p.res = nullptr; p.res_scale = 1.f; p.out = reinterpret_cast<bf16*>(out); p.stats = stats;
    p.M = M; p.HW = HW; p.Cout = 64; p.Cout_pad = 64; p.nch = 2;
    if (elem == 2) return run_g1<64, XF_NONE, 0, bf16, 2, true, true>(p, s);
    return run_g1<64, XF_NONE, 0, f16, 2, false, true>(p, s);
}

}  // namespace hsidm
