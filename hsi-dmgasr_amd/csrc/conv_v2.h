// conv_v2: persistent implicit-GEMM 3x3 convolution (stride 1, optional nearest-x2 input) for the
// bf16 throughput mode.  Successor of conv_igemm.h for the shapes that carry ~90 % of the UNet's time.
//
// What changed against v1, and the measurements that motivated it (profiles/r01_baseline, r01_pmc):
//   v1: 15 VALU + 5 SALU instructions per MFMA, waves parked 43 % of the time (SQ_WAIT_ANY) behind one
//   barrier + one LDS weight-tile commit per K step; MFMA pipe 15 % busy.
//   * weights never touch LDS: each wave owns a 32-wide cout slice and streams its MFMA B-fragments
//     straight from L2 into registers (pre-packed so that one wave-load is one contiguous 1 KiB),
//     through a 3-deep register ring issued two K steps ahead -> no per-step barrier, no LDS writes;
//   * the input halo tile is double-buffered in LDS; the 9 taps of a chunk are fully unrolled with a
//     STATIC schedule: tap i issues the load of the next chunk's i-th halo vector, tap i+3 normalises /
//     SiLUs / packs it into the other buffer in the shadow of that tap's MFMAs
//     (nothing is consumed right after its load, so no wait drains the weight ring); one LDS-only
//     barrier per chunk; A fragments are double-buffered across the four k-slices of a tap;
//   * SiLU uses v_exp + v_rcp on u = log2(e)*y, which the GroupNorm table delivers pre-scaled (silu_log2e: 4 VALU per
//     element after the affine map instead of an IEEE division; the factor is removed from the accumulators);
//   * the workgroup is persistent: it walks (pixel tile, cout slice) work items round-robin so the
//     weight ring and the halo prefetch run across tile boundaries; a block always works on the same
//     cout slice, and blocks sharing an XCD (blockIdx % 8) share that slice -> each L2 streams
//     1/Nslices of the weights;
//   * the epilogue emits per-(image, tile part, channel) sum / sum-of-squares of what it stored, so
//     the consumer's GroupNorm needs no pass over the tensor (hsidm_gn_finalize reads the slab).
//
// Tile: 128 output pixels (8x16 of one image, or 8x8 of two) x BN couts, 4 waves; wave tile =
// (128/WM) pixels x 32 couts, WM x WN = 4, WN = BN/32.  K step = 64 input channels of one tap.
//
// Matrix instruction (round 3): v_mfma_f32_16x16x32.  It costs the same cycles per FLOP as the 32x32x16 form, but under the chip's
// power management the clock held on it is higher (tools/ubench/mfma_shape.hip, a loop shaped like this one: 1 310 vs 1 150 TFLOP/s;
// MI355X_MICROARCH.md "DVFS give-back" item 7) - and these kernels sit at that limit.  A 32-pixel x 32-cout block is four of them:
// (16-pixel half r) x (16-cout half nh).  A operand of half r: lane (c = lane % 16, g = lane / 16) reads channels 8g .. 8g+7 of a
// 32-channel slice of pixel c (one tile row of 16, or two of 8); pixels are 160 bytes apart and the row pitch of the 8-wide tile is
// 1792 bytes, which makes the four 16-lane service groups of every ds_read_b128 cover all 64 banks once at every tap.  B operand
// (nh, slice q): cout 16 nh + c, channels 32q + 8g .. +7 - a 16-byte unit of the SAME packed weight order as before (fragment
// 2q + g/2, lane 32 (g%2) + 16 nh + c): no host-side change.  Sub-steps of a tap: kk = 2q + r, each 2 MR MFMAs (both cout halves).
#pragma once
#include "conv_igemm.h"

namespace hsidm {

struct ConvV2Params {
    const bf16* src0;       // NHWC [B][Hin][Win][C0]
    const bf16* src1;       // NHWC [B][Hin][Win][C1] or null (channel concat)
    const f32x4* gn_ab;     // [B][Ctot/2] = (scale, shift) pairs of two channels, or null
    int C0, C1, nchunks;
    // fused 1x1 projection (PROJ kernels; hsidm_conv_desc.ph[1]): pchunks more 64-channel chunks of one tap, read raw from (psrc0 | psrc1)
    const bf16* psrc0;
    const bf16* psrc1;
    int PC0, PC1, pchunks;
    const bf16* w;          // packed [step][Cout_pad/32][kk 4][lane 64][8], step = chunk*9 + tap
    const bf16* w_lo;       // NP = 2 kernels: the low halves of the weights (w + w_lo: an absolute granularity of 6e-8 - the low halves are fp16 subnormals - i.e. ~18-19 significant bits at |w| ~ 0.03), same layout; else null
    const bf16* w_ls;       // sparse-lo kernels (conv_v3.hip, SPL): the low halves 2:4-compressed, [step][Cout_pad/32][2][64][8] (include/hsidm.h, w_v2_ls)
    const int* w_li;        // ... and their index words [step][Cout_pad/32][64]
    const float* bias;
    const float* film;
    int film_stride;
    const bf16* res;
    float res_scale;
    bf16* out;
    float2* stats;          // [B][tiles_per_img*SUBS][Cout] or null
    int B, Hin, Win, Hout, Wout, Cout, Cout_pad;
    int ups, act;
    int tiles_x, tiles_y;
    int m_tiles, n_slices, total_items, steps_per_item;
    int up_m;               // UP4 kernels: XCDs per cout slice when the L2-friendly (tile, parity) order applies, else 0
    int xcd_m;              // other kernels: XCDs per cout slice when each of them walks a contiguous range of pixel tiles, else 0
    int tpi_shift, tx_shift; // log2 of tiles per image / per tile row when those are powers of two, else -1
    int ns_shift, xm_shift;  // conv_v2: log2 of n_slices / of the XCD group count (xcd_m or up_m: 8 / n_slices) when powers of two, else -1
    unsigned long long* stamps;   // diagnostic build (HSIDM_V2_STAMPS): [block][wave][item<8][slot<16] s_memtime
    int abl;                // diagnostic ablation mask (HSIDM_V2_ABL): 1 no stores, 2 no transform, 4 no halo loads, 8 no weight loads, 16 no commits
};

// UP4: nearest-x2 upsample folded into the weights.  Output pixel (2y+py, 2x+px) of conv3x3(nearest_x2(in)) only sees
// the 2x2 input neighbourhood {y-1+py, y+py} x {x-1+px, x+px}; the host pre-sums the 3x3 taps that fall on the same input
// pixel into four 2x2 kernels (one per output parity).  A work item is (input tile, parity, cout slice): K = 4*Cin per
// output pixel instead of 9*Cin, and the halo tile is the plain input tile (each input pixel staged once per parity
// instead of ~3.3 times per output tile).
//
// DN4 (SP_ = 2): the stride-2 conv on the same 2x2-window schedule.  Output (y, x) reads input rows 2y-1, 2y, 2y+1: the odd
// rows 2(y-1)+1 and 2y+1 are rows y-1 and y of the odd-row plane, the even row is row y of the even-row plane (same for
// columns).  The K loop walks the four (row parity, column parity) planes of the input; a staged halo pixel (iy, ix) of
// plane (ry, rx) is in[2*iy + ry][2*ix + rx] and every plane contributes the window {iy-1, iy} x {ix-1, ix} with zero
// weights where the plane has no tap (16/9 of the minimal MFMAs, on convs that are 1 % of the step's FLOPs).
// E_: element type of activations, weights and MFMA operands (bf16 | fp16: common.h, Elem).
// NP_ = 2: every MFMA is issued twice, on the high and on the low half of the weight (w_lo): the weights then carry ~22
// significant bits at the price of twice the matrix instructions - affordable on the layers that are bound by the staging
// transform, not by the matrix pipe (the 64- and 128-cout layers of the two high-resolution levels), where it removes the one
// SYSTEMATIC error of a 16-bit mode (the same weight rounding in every step of the chain).  The weight ring
// is then fragment-granular (6 pairs, five k-slices ahead) instead of three K steps deep: the same registers.
//
// S_ = float, AP_ = 2 (with E_ = bf16, NP_ = 2): the fp32 mode on this kernel.  Activations are stored in fp32 (two 16-byte vectors per
// staged pixel-vector, fp32 GroupNorm pairs, a 4-byte epilogue patch, fp32 stores); every staged value is split into bf16 hi + lo
// into TWO halo tiles per buffer and every product is three MFMAs, lo*hi + hi*lo + hi*hi (conv_igemm.h's arithmetic, ~2^-17 per
// product).  Four halo tiles are 113 KB: one workgroup per CU, up to 512 registers per wave.
//
// SPL_ (fp16, NP_ = 2; the one-image 8x16 tile on four waves: stride-1 3x3 and the folded up / down-sampling forms): the second weight pass on v_smfmac_f32_16x16x64_f16 with the low
// halves 2:4-compressed (ConvV2Params::w_ls / w_li; conv_v3.hip has the same form and the reasoning): the weights are the A operand of
// both passes, the accumulators come out transposed (a lane holds couts 16 nh + 4 g .. + 3 of one pixel), sub-steps run in the order
// (tap, 16-pixel half r, 32-channel slice q) with the tap's dense fragments held for both halves - two register sets alternating
// between taps, one tap of lookahead - and one sparse instruction per (r, 32-pixel group, cout half) after slice q = 1.
template <int BN_, int TH_, int TW_, int NI_, int XF_, int SP_ = 0, int NW_ = 4, typename E_ = bf16, int NP_ = 1, typename S_ = E_, int AP_ = 1,
          bool SPL_ = false>
struct V2Cfg {
    using E = E_;
    using S = S_;                                               // storage type of activations in HBM
    static constexpr bool SPL = SPL_;
    static_assert(!SPL_ || (NP_ == 2 && AP_ == 1 && sizeof(S_) == 2 && NI_ == 1 && NW_ == 4 && TW_ == 16), "sparse low halves: see above");
    static constexpr int NP = NP_, AP = AP_;
    static constexpr bool F32 = sizeof(S_) == 4;
    static constexpr int SV = F32 ? 2 : 1;                      // 16-byte vectors per staged 8-channel pixel-vector
    static_assert(NP_ == 1 || NP_ == 2, "weight passes");
    static_assert(AP_ == 1 || (AP_ == 2 && NP_ == 2 && F32), "split activations come with split weights and fp32 storage");
    static constexpr int BN = BN_, TH = TH_, TW = TW_, NI = NI_, XF = XF_;
    // NW = 8: 512 threads, one workgroup per CU, 256 couts per item - the eight waves share one staged (transformed) halo tile,
    // so the GroupNorm+SiLU transform is paid once per 256 couts instead of once per 128
    static constexpr int NW = NW_, NTHR = 64 * NW_;
    static constexpr bool UP4 = SP_ == 1, DN4 = SP_ == 2;
    static constexpr bool FR = SP_ != 0;                        // 4-tap chunks, fragment-granular weight ring
    static constexpr int NT = FR ? 4 : 9;                       // taps per channel chunk
    static constexpr bool FRG = (FR || NP_ == 2) && !SPL_;      // weights through the fragment ring (else the 3-step ring; SPL: tap sets)
    // ring slots / lookahead in k-slices; the slot count divides the k-slices of a chunk (16 | 36) so that every index is static
    // (a 32-channel slice keeps its two fragments through both of its sub-steps: FL <= FS - 2)
    // (the fp32 form has one workgroup per CU and registers to spare: a deeper ring, ten / six fragments ahead)
    // (folded forms: 8 slots, six ahead, with hi + lo weights too - 4 / 2 kept their launches 7 - 12 % longer: up 128 -> 128 968 -> 903 us)
    static constexpr int FS = AP_ == 2 ? (FR ? 8 : 12) : (FR ? 8 : 6), FL = AP_ == 2 ? FS - 2 : (FR ? 6 : 4);
    static constexpr int BM = TH * TW * NI, BK = 64;           // 128 pixels; 64 for the one-image 8x8 tile (maps of 8x8 pixels
                                                                // at batches too small to fill the GPU with two-image tiles)
    static_assert(BM == 128 || (BM == 64 && BN_ == 128), "tile");
    static constexpr int WN = BN / 32, WM = NW / WN, MR = BM / WM / 32;
    static constexpr int HROWS = TH + 2, HCOLS = TW + 2, HPIX = HROWS * HCOLS;
    static constexpr int PSTR = BK + 16, VPP = BK / 8;          // 160-byte pixels (the last 32 B are padding; see the bank note above)
    static constexpr int HVEC = NI * HPIX * VPP;
    static constexpr int MAXHV = (HVEC + NTHR - 1) / NTHR;
    static constexpr int NH = FR ? MAXHV : 4;                   // raw staging registers (vectors in flight)
    static_assert(MAXHV <= 7, "one staged vector per tap 2..8");
    static_assert(!FR || XF_ == 0, "the up/downsample convs have no GroupNorm prologue");
    // LDS pitch of one halo row (elements).  TW = 16: an A operand is one tile row, any pitch serves; TW = 8: it is two tile rows
    // of 8, conflict-free at 112 sixteen-byte units (searched: the only pitch in [100, 116] units that is, with 10-unit pixels).
    static constexpr int RP = (TW == 16) ? HCOLS * PSTR : 896;
    static_assert(RP >= HCOLS * PSTR, "row pitch");
    static constexpr int HALO_ELEMS = NI * HROWS * RP;
    // two halo buffers + the per-thread table of halo positions (MAXHV packed ints per thread, see describe)
    static constexpr size_t POS_BYTES = (size_t)MAXHV * NTHR * 4;
    // NW = 8: the epilogue's transposition patches (8 x 5 KiB) do not fit the free halo buffer: a region of their own
    static_assert(!(AP_ == 2 && NW_ == 8), "the 8-wave form has no room for a second halo tile");
    // The epilogue's transposition patches (one per wave: 64 pixels x 40 elements of the storage type) live in the halo buffer the
    // item has just freed - unless that buffer is too small for them (the one-image 8x8 tile: 16.6 KB per tile against 20 KB of
    // patches, which used to run over the position table behind the buffers; the fp32 form of that tile: 33 KB against 40 KB) or
    // the workgroup has 8 waves: then they get a region of their own behind the position table.
    // (EP = 32-pixel MFMA row groups per epilogue pass: two - except in the fp32-storage form with ONE activation operand (AP = 1, the
    // "fp32h" kernel set), whose 4-byte patches of 64 pixels would not fit the freed halo buffer, and a region of their own would not
    // leave room for two workgroups per CU: it transposes 32 pixels per pass, 20 KB for the four waves)
    static constexpr int EP = (F32 && AP_ == 1) ? 1 : 2;
    static constexpr size_t PATCH_BYTES = (size_t)NW * (32 * EP) * 40 * sizeof(S_);
    static constexpr bool OWN_PATCH = NW == 8 || PATCH_BYTES > (size_t)AP * HALO_ELEMS * 2;
    static constexpr size_t LDS_BYTES = (size_t)2 * AP * HALO_ELEMS * 2 + POS_BYTES + (OWN_PATCH ? PATCH_BYTES : 0);
    // depth of the A-operand ring in sub-steps (kernel: a_fetch).  Two where a sub-step carries twice the MFMAs (NP = 2) and on the
    // two-image 128-cout GroupNorm form, where the third set made the allocator spill 37 registers (per-image FiLM / statistics
    // registers on top of MR = 4)
    static constexpr int AD = (SPL_ || NP_ == 2 || (NI_ == 2 && XF_ != 0 && BN_ == 128 && NW_ == 4)) ? 2 : 3;
    // statistics sub-entries per spatial tile and image (see epilogue)
    static constexpr int SUBS = (NI == 1) ? WM : (WM >= 2 ? WM / 2 : 1);
};

template <int V> struct SlotTag { static constexpr int value = V; };

#ifdef HSIDM_V2_STAMPS
#define HSIDM_STAMP(it_, slot_)                                                                             \
    do {                                                                                                     \
        if (p.stamps && (it_) < 8 && lane == 0) {                                                            \
            unsigned long long t_;                                                                           \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
            p.stamps[(((size_t)blockIdx.x * 4 + wave) * 8 + (it_)) * 16 + (slot_)] = t_;                     \
        }                                                                                                    \
    } while (0)
#else
#define HSIDM_STAMP(it_, slot_) do {} while (0)
#endif

// The K loop runs at wave priority 1, the epilogue at 0: the two workgroups of a CU are out of phase, and the one in its
// (VALU-heavy) epilogue otherwise wins issue slots by age from the one feeding the matrix pipe.  Measured in one box, six
// alternating runs: 12.67 vs 12.73 ms per step (+0.45 %; the inverse assignment is neutral).  -DHSIDM_NO_PRIO removes it.
#ifndef HSIDM_NO_PRIO
#define HSIDM_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define HSIDM_SETPRIO(n) do {} while (0)
#endif

#ifdef HSIDM_V2_ABLATE
#define HSIDM_ABL(mask) (p.abl & (mask))
#else
#define HSIDM_ABL(mask) false
#endif

__device__ __forceinline__ void lds_barrier() {
    // LDS-only ordering: keeps the register-ring weight loads in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    h16x2 h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float h2_lo(unsigned u) { return (float)__builtin_bit_cast(h16x2, u)[0]; }
__device__ __forceinline__ float h2_hi(unsigned u) { return (float)__builtin_bit_cast(h16x2, u)[1]; }

__device__ __forceinline__ float silu_fast(float y) {
    // y * sigmoid(y) with v_exp_f32 / v_rcp_f32 (about 1e-6 relative; the result is rounded to bf16)
    return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
}
// The staging transform works on u = log2(e) * y (the GroupNorm table's third part is pre-multiplied): u / (1 + 2^-u) =
// log2(e) * y * sigmoid(y) without the multiplication that turns y into an exp2 argument - one VALU instruction less per
// staged element, in kernels bound by VALU issue.  The factor leaves on the fp32 accumulators (epilogue: fma(acc, ln 2, bias)).
constexpr float kLn2 = 0.693147181f;
// two fp32 -> one packed bf16 pair (v_cvt_pk_bf16_f32); storing the halves separately uses ds_write_b16 / ds_write_b16_d16_hi,
// i.e. one conversion per two epilogue values instead of one each
__device__ __forceinline__ float silu_log2e(float u) {
    return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u));
}

// Lane id recomputed where it is needed.  `volatile` keeps the two instructions in place: the builtin form is loop-invariant,
// so the compiler hoisted it, then spilled the result and paid an s_waitcnt vmcnt(0) for every reload.
__device__ __forceinline__ int lane_id_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// value of lane (l ^ M).  M < 32: ds_swizzle in bit-mask mode (the pattern is an immediate: no address register; the address
// registers of __shfl_xor were hoisted out of the item loop, spilled, and reloaded behind the epilogue's stores).  M = 32: one
// ds_bpermute with the address built from the caller's (freshly recomputed) lane id.
template <int M>
__device__ __forceinline__ float lane_xor(float v, int lane_now) {
    if (M < 32) return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (M << 10) | 0x1f));
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane_now ^ 32) << 2, __float_as_int(v)));
}

template <typename C>
__global__ __launch_bounds__(C::NTHR, (C::NW == 8 || C::AP == 2) ? 1 : 2) void conv_v2_kernel(const ConvV2Params p) {
    constexpr int BN = C::BN, TH = C::TH, TW = C::TW, NI = C::NI, MR = C::MR, WN = C::WN, WM = C::WM;
    constexpr int HPIX = C::HPIX, HCOLS = C::HCOLS, PSTR = C::PSTR, VPP = C::VPP, BK = C::BK, RP = C::RP, HROWS = C::HROWS;
    constexpr int MAXHV = C::MAXHV, NT = C::NT, NH = C::NH;
    constexpr bool UP4 = C::UP4, DN4 = C::DN4, FR = C::FR;
    constexpr int NTHR = C::NTHR;
    constexpr int NP = C::NP, FS = C::FS, FL = C::FL;
    constexpr bool FRG = C::FRG, SPL = C::SPL;
    using E = typename C::E;
    using S = typename C::S;
    constexpr int AP = C::AP, SV = C::SV;
    constexpr bool F32 = C::F32;
    constexpr int BUFE = AP * C::HALO_ELEMS;                   // elements of one halo buffer (hi tile [+ lo tile])
    using EL = Elem<E>;
    using x8 = typename EL::x8;
    using x2 = typename EL::x2;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* halo = reinterpret_cast<E*>(smem_raw);                // [2][HALO_ELEMS]
    if constexpr (sizeof(E) == 2 && !__is_same(E, bf16)) fp16_saturating_stores();      // (common.h)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WM == 1 ? 0 : wave / WN, wn = wave % WN;    // (WM == 1: a constant, so that per-image arrays are indexed statically)
    const int lc = lane & 15, lg = lane >> 4;                  // MFMA 16x16x32: row / column lc, k group lg
    const int G = gridDim.x;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ctot = p.C0 + p.C1;

    int item = blockIdx.x;                                     // item = m_tile * n_slices + n_slice
    const int ns = item % p.n_slices;                          // constant for this block (G % n_slices == 0)
    const int n0 = ns * BN;
    const int n_items_blk = (p.total_items - item + G - 1) / G;

    // ---- weight stream: one contiguous 1 KiB per (step, 32-cout slice, kk) --------------------------------
    const int nsw = p.Cout_pad >> 5;
    // (per lane: the 16-byte unit (cout 16 nh + lc, channels 32q + 8 lg ..) of fragment e = 2q + nh sits at frag_off(e) behind this)
    const size_t wlane_off = ((size_t)(ns * WN + wn) * 4 * 64 + (lg >> 1) * 64 + 32 * (lg & 1) + lc) * 8;
    auto frag_off = [](int e) __attribute__((always_inline)) -> int { return (e >> 1) * (2 * 64 * 8) + (e & 1) * (16 * 8); };
    const E* wlane = reinterpret_cast<const E*>(p.w) + wlane_off;
    const E* wlane_lo = reinterpret_cast<const E*>(NP == 2 ? p.w_lo : p.w) + wlane_off;
    const size_t wstep_stride = (size_t)nsw * 4 * 64 * 8;
    x8 ring[(FRG || SPL) ? 1 : 3][4];
    // SPL: register sets (tap & 1): four dense fragments e = 2q + nh, the two sparse halves, one index word (low / high half = cout half)
    x8 whi[SPL ? 8 : 1], wls[SPL ? 4 : 1];
    int wli[SPL ? 2 : 1];
    const E* lsp = reinterpret_cast<const E*>(SPL ? p.w_ls : p.w) + ((size_t)(ns * WN + wn) * 2 * 64 + lane) * 8;
    const int* lip = (SPL ? p.w_li : reinterpret_cast<const int*>(p.w)) + (size_t)(ns * WN + wn) * 64 + lane;
    // item -> (pixel tile, output parity).  UP4 with up_m > 0: the four parities of one input tile are consecutive work
    // of ONE XCD (blocks b, b+8, b+16, b+24), so its L2 fetches the tile once.
    // (divisions by run-time values are ~20 scalar instructions each and an item needs some twenty of them: shifts whenever the
    // divisor is a power of two, which the UNet's tile grids, slice counts and XCD groups are)
    auto div_ns = [&](int x) __attribute__((always_inline)) -> int { return p.ns_shift >= 0 ? x >> p.ns_shift : x / p.n_slices; };
    auto div_tpi = [&](int x) __attribute__((always_inline)) -> int { return p.tpi_shift >= 0 ? x >> p.tpi_shift : x / (p.tiles_x * p.tiles_y); };
    auto div_tx = [&](int x) __attribute__((always_inline)) -> int { return p.tx_shift >= 0 ? x >> p.tx_shift : x / p.tiles_x; };
    auto item_tile = [&](int it, int& par) __attribute__((always_inline)) -> int {
        const int mt = div_ns(it);
        // blocks of one XCD (b % 8) that share a cout slice hold tiles mt = x, x + m, x + 2m, ... (m = 8 / n_slices); re-ordered so
        // that each XCD walks a contiguous range, neighbouring tiles - which share halo rows and columns - meet in one L2
        if (!UP4) {
            par = 0;
            if (p.xcd_m <= 0) return mt;
            if (p.xm_shift >= 0) return (mt & (p.xcd_m - 1)) * (p.m_tiles >> p.xm_shift) + (mt >> p.xm_shift);
            return (mt % p.xcd_m) * (p.m_tiles / p.xcd_m) + mt / p.xcd_m;
        }
        if (p.up_m > 0) {
            if (p.xm_shift >= 0) { par = (mt >> p.xm_shift) & 3; return ((mt >> (p.xm_shift + 2)) << p.xm_shift) + (mt & (p.up_m - 1)); }
            par = (mt / p.up_m) & 3;
            return (mt / (4 * p.up_m)) * p.up_m + mt % p.up_m;
        }
        par = mt & 3;
        return mt >> 2;
    };
    int wnext = 0;                                              // step (within an item) of the next weight fetch
    int w_base = 0, w_base_nx = 0;                              // UP4: first weight step of the parity of the item the ring fetches for /
                                                                // of the item after it (set at every item start: no division in f_issue)
    if (UP4) { int par; item_tile(item, par); w_base = par * p.steps_per_item; }
    auto b_issue = [&](x8 (&dst)[4]) __attribute__((always_inline)) {
        const E* src = wlane + (size_t)wnext * wstep_stride;
        if (!(HSIDM_ABL(8))) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dst[kk] = *reinterpret_cast<const x8*>(src + frag_off(kk));
        }
        wnext = (wnext + 1 == p.steps_per_item) ? 0 : wnext + 1;
    };
    // UP4: 16 fragments per chunk do not keep a 3-step ring in phase (4 taps), so the ring holds single fragments:
    // 8 slots, fetched six sub-steps (24 MFMAs) ahead.  NP = 2: (high, low) fragment pairs, FS slots, FL sub-steps ahead.
    x8 fring[FRG ? FS : 1], fring_lo[(FRG && NP == 2) ? FS : 1];
    auto f_issue = [&](int slot, int kk) __attribute__((always_inline)) {
        if (!(HSIDM_ABL(8))) {
            const size_t off = (size_t)(w_base + wnext) * wstep_stride + frag_off(kk);
            fring[FRG ? slot : 0] = *reinterpret_cast<const x8*>(wlane + off);
            if (NP == 2) fring_lo[NP == 2 ? slot : 0] = *reinterpret_cast<const x8*>(wlane_lo + off);
        }
        if (kk == 3) {
            if (wnext + 1 == p.steps_per_item) {
                wnext = 0;
                w_base = w_base_nx;
            } else wnext += 1;
        }
    };

    auto s_issue = [&](int set, int part) __attribute__((always_inline)) {      // SPL: part 0..3 of weight step (w_base +) wnext into register set `set`
        const size_t st = (size_t)(w_base + wnext);
        whi[SPL ? set * 4 + part : 0] = *reinterpret_cast<const x8*>(wlane + st * wstep_stride + frag_off(part));
        if (part < 2) wls[SPL ? set * 2 + part : 0] = *reinterpret_cast<const x8*>(lsp + (st * nsw * 2 + part) * (64 * 8));
        if (part == 2) wli[SPL ? set : 0] = lip[st * nsw * 64];
        if (part == 3) {
            if (wnext + 1 == p.steps_per_item) {
                wnext = 0;
                w_base = w_base_nx;                             // (UP4: the next item's parity)
            } else wnext += 1;
        }
    };

    // ---- halo staging state ---------------------------------------------------------------------------------
    const int cv = tid % VPP;
    auto hv_lds = [&](int i) __attribute__((always_inline)) -> int {   // LDS offset of staged vector i (halo pixel tid/VPP + i*256/VPP)
        const int hp = tid / VPP + i * (NTHR / VPP);
        const int row = hp / HCOLS;                            // (image, halo row) pairs are consecutive rows of the LDS image
        return row * RP + (hp - row * HCOLS) * PSTR + cv * 8;
    };
    const bool last_live = tid + (MAXHV - 1) * NTHR < C::HVEC;  // the last vector slot is partial
    int hv_pix[MAXHV];                                          // for the tile being STAGED
    auto hv_pos = [&](int i, int t) __attribute__((always_inline)) -> int {   // img<<16 | hy<<8 | hx, or -1 (dead slot); recomputed, not kept
        const int hp = t / VPP + i * (NTHR / VPP);
        const int img = hp / HPIX;
        const int r = hp - img * HPIX;
        const int hy = r / HCOLS, hx = r - hy * HCOLS;
        return (i < MAXHV - 1 || last_live) ? ((img << 16) | (hy << 8) | hx) : -1;
    };
    // The positions are constant per thread; recomputing them per item cost ~25 VALU per vector (two divisions by
    // constants), keeping them in registers made the allocator spill.  They live in LDS instead: one ds_read per vector.
    int* pos_tab = reinterpret_cast<int*>(smem_raw + (size_t)2 * BUFE * 2);
#pragma unroll
    for (int i = 0; i < MAXHV; ++i) pos_tab[i * NTHR + tid] = hv_pos(i, tid);
    int st_b0 = 0;
    auto tile_coords = [&](int it, int& b0, int& oy0, int& ox0) __attribute__((always_inline)) {   // tile origin on the staged grid
        int par_;
        const int mt = item_tile(it, par_);
        const int tg = div_tpi(mt);
        const int tr = mt - tg * tiles_per_img;
        b0 = tg * NI;
        const int try_ = div_tx(tr);
        oy0 = try_ * TH;
        ox0 = (tr - try_ * p.tiles_x) * TW;
    };
    auto describe = [&](int it) __attribute__((always_inline)) {
        int b0, oy0, ox0;
        tile_coords(it, b0, oy0, ox0);
        st_b0 = b0;
        const bool up = !FR && p.ups;
        const int hlim = DN4 ? p.Hout : (up ? 2 * p.Hin : p.Hin), wlim = DN4 ? p.Wout : (up ? 2 * p.Win : p.Win);
        const int sh = up ? 1 : 0;
        int posv[MAXHV];                                        // all table reads first: one LDS round trip instead of MAXHV
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) posv[i] = pos_tab[i * NTHR + tid];
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) {
            const int pos = posv[i];
            const int b = b0 + (pos >> 16);
            const int iy = oy0 + ((pos >> 8) & 255) - 1, ix = ox0 + (pos & 255) - 1;
            const bool ok = pos >= 0 && b < p.B && (unsigned)iy < (unsigned)hlim && (unsigned)ix < (unsigned)wlim;
            hv_pix[i] = !ok ? -1 : (DN4 ? (b * p.Hin + 2 * iy) * p.Win + 2 * ix : (b * p.Hin + (iy >> sh)) * p.Win + (ix >> sh));
        }
    };
    u32x4 hreg[NH][SV];            // staged raw vectors: vector i lives in slot i % NH from its issue tap to its commit tap
    unsigned abh[8];               // (scale, shift) of this thread's 8 channels, packed fp16x2: 11-bit significands, the
                                   // transformed value is rounded to bf16 (8 bits) anyway; halves the registers held across taps
    float gsc[F32 ? 8 : 1], gsh[F32 ? 8 : 1];   // fp32 mode: the fp32 pairs (first part of the GroupNorm table)
    bool st_cok = true;
    int st_c = 0, st_cs = 0, st_cl = 0, st_plane = 0;
    // (scale, shift) of 8 consecutive channels of image b: two 16-byte loads from the fp16x2 half of the GroupNorm table
    // (hsidm_gn_finalize writes it behind the fp32 pairs), consumed at commit time: nothing waits on them when they are requested
    auto gn_params = [&](int b, int c) __attribute__((always_inline)) {
        if constexpr (F32) {
            const f32x4* t = p.gn_ab + (((size_t)b * ctot + c) >> 1);          // [B][C] (scale, shift) pairs, two channels per f32x4
            f32x4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = t[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) { gsc[2 * k] = r[k][0]; gsh[2 * k] = r[k][1]; gsc[2 * k + 1] = r[k][2]; gsh[2 * k + 1] = r[k][3]; }
            return;
        }
        const u32x4* t = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned*>(p.gn_ab) + (size_t)3 * p.B * ctot + (size_t)b * ctot + c);
        const u32x4 lo = t[0], hi = t[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) { abh[k] = lo[k]; abh[4 + k] = hi[k]; }
    };
    const S* st_src = reinterpret_cast<const S*>(p.src0);
    auto halo_begin = [&](int chunk) __attribute__((always_inline)) {
        // channel slice of the chunk being staged (+ its GroupNorm parameters).  Loads are UNCONDITIONAL (clamped
        // addresses): a predicated load would be merged with its zero alternative right away and that use would
        // wait for the load, draining the weight ring; out-of-range data is zeroed at commit time instead.
        int c = chunk * BK + cv * 8;
        if (DN4) {                                              // chunk = (input plane, channel chunk)
            const int nch_c = p.nchunks >> 2;
            const int plane = chunk / nch_c;
            c = (chunk - plane * nch_c) * BK + cv * 8;
            st_plane = (plane >> 1) * p.Win + (plane & 1);
        }
        st_cok = c < ctot;
        const int cc = st_cok ? c : 0;
        if (cc < p.C0) { st_src = reinterpret_cast<const S*>(p.src0); st_cs = p.C0; st_cl = cc; }
        else           { st_src = reinterpret_cast<const S*>(p.src1); st_cs = p.C1; st_cl = cc - p.C0; }
        st_c = cc;
        if (C::XF != XF_NONE && NI == 1) gn_params(st_b0, cc);
    };
    auto halo_issue_one = [&](int i) __attribute__((always_inline)) {
        if (HSIDM_ABL(4)) return;
        const int pix = hv_pix[i] >= 0 ? hv_pix[i] + (DN4 ? st_plane : 0) : 0;
        const u32x4* src_v = reinterpret_cast<const u32x4*>(st_src + (size_t)pix * st_cs + st_cl);
#pragma unroll
        for (int h = 0; h < SV; ++h) hreg[i % NH][h] = src_v[h];
    };
    // dead slot of the last (partial) vector round: the store goes to the row padding instead of being branched around,
    // so that the commit stays in the MFMAs' basic block and the scheduler can interleave the two
    const int dead_off = ((tid % (NI * HPIX)) / HCOLS) * RP + ((tid % (NI * HPIX)) % HCOLS) * PSTR + BK + 8 * ((tid / (NI * HPIX)) & 1);   // a pixel's padding
    // The commit of one staged vector is cut into four slices (two elements each) that ride in the four k-slices of a tap,
    // so that every scheduling region holds 4 MFMAs and ~1/4 of the transform instead of 16 MFMAs followed by a block of
    // 70 VALU instructions (a single wave then overlaps the two; measured with one workgroup per CU).
    float cm_v[8];
    auto halo_commit_part = [&](int i, int buf, int part) __attribute__((always_inline)) {
        if (part == 0 && C::XF != XF_NONE && NI > 1 && !(HSIDM_ABL(2))) {
            const int img = (tid / VPP + i * (NTHR / VPP)) / HPIX;
            const int bb = (st_b0 + img < p.B) ? st_b0 + img : st_b0;
            gn_params(bb, st_c);
        }
        {
            if constexpr (F32) {
                cm_v[2 * part] = __uint_as_float(hreg[i % NH][part >> 1][(part & 1) * 2]);
                cm_v[2 * part + 1] = __uint_as_float(hreg[i % NH][part >> 1][(part & 1) * 2 + 1]);
                if (C::XF != XF_NONE && !(HSIDM_ABL(2))) {
#pragma unroll
                    for (int k = 2 * part; k < 2 * part + 2; ++k) cm_v[k] = silu_fast(fmaf(cm_v[k], gsc[k], gsh[k]));
                }
            } else {
                const unsigned w = hreg[i % NH][0][part];
                cm_v[2 * part] = EL::lo(w);
                cm_v[2 * part + 1] = EL::hi(w);
                if (C::XF != XF_NONE && !(HSIDM_ABL(2))) {
#pragma unroll
                    for (int k = 2 * part; k < 2 * part + 2; ++k) cm_v[k] = silu_log2e(fmaf(cm_v[k], h2_lo(abh[k]), h2_hi(abh[k])));
                }
            }
        }
        if (part == 3) {
            const bool live = st_cok && hv_pix[i] >= 0;         // zero padding stays zero (pad AFTER activation)
            x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (E)cm_v[k];
            u32x4 ou = __builtin_bit_cast(u32x4, o);            // zero padding / dead channels: select on the packed words
#pragma unroll
            for (int k = 0; k < 4; ++k) ou[k] = live ? ou[k] : 0u;
            const int off = (i == MAXHV - 1 && !last_live) ? dead_off : hv_lds(i);
            *reinterpret_cast<u32x4*>(halo + buf * BUFE + off) = ou;
            if constexpr (AP == 2) {                            // low halves x - bf16(x) into the buffer's second tile
                x8 ol;
#pragma unroll
                for (int k = 0; k < 8; ++k) ol[k] = (E)(cm_v[k] - (float)o[k]);
                u32x4 olu = __builtin_bit_cast(u32x4, ol);
#pragma unroll
                for (int k = 0; k < 4; ++k) olu[k] = live ? olu[k] : 0u;
                *reinterpret_cast<u32x4*>(halo + buf * BUFE + C::HALO_ELEMS + off) = olu;
            }
        }
    };
    auto halo_commit_one = [&](int i, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 4; ++part) halo_commit_part(i, buf, part);
    };

    // ---- MFMA fragment bases --------------------------------------------------------------------------------------
    int abase[MR];                                              // 16-pixel half r = 0 of the 32-pixel group mr; half 1 is RHALF further
    constexpr int RHALF = (16 / TW) * RP;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int pm = wm * (C::BM / WM) + mr * 32 + lc;
        const int img = pm / (TH * TW);
        const int q = pm - img * (TH * TW);
        const int ty = q / TW, tx = q - ty * TW;
        abase[mr] = (img * HROWS + ty) * RP + tx * PSTR + 8 * lg;
    }
    f32x4 acc[MR][2][2];                                        // [32-pixel group][16-pixel half][16-cout half]

    // ---- staging cursor: the (item, chunk) whose halo tile is fetched next -------------------------------------------
    const int nch = p.nchunks;
    int cur = 0;                 // halo buffer being consumed
    int st_item = item, st_chunk = 0;
    bool st_valid = true;
    auto stage_advance = [&]() __attribute__((always_inline)) {
        if (st_chunk + 1 < nch) { st_chunk += 1; }
        else { st_chunk = 0; st_item += G; }
        st_valid = st_item < p.total_items;
    };

    // prologue: first two weight steps, first halo tile (synchronously)
    if constexpr (SPL) {
#pragma unroll
        for (int f = 0; f < 4; ++f) s_issue(0, f);
    } else if (!FRG) {
        b_issue(ring[0]);
        b_issue(ring[1]);
    } else {
#pragma unroll
        for (int f = 0; f < FL; ++f) f_issue(f, f % 4);
    }
    describe(item);
    halo_begin(0);
#pragma unroll
    for (int i = 0; i < MAXHV; ++i) {
        halo_issue_one(i);
        halo_commit_one(i, 0);
    }
    lds_barrier();
    stage_advance();

    // Per-lane output channels; bias[n] + FiLM[image][n] are added to the accumulators in the epilogue.  They are loaded with
    // ordinary (compiler-tracked) loads at the first tap of the item's LAST chunk: vmcnt retires in order and that chunk issues 36+
    // weight loads behind them, so the counted waits on those weights already cover them - the ISA shows no additional s_waitcnt.
    // (Until round 3 they were fetched at the start of the item with loads the compiler did not track - as a tracked value that is
    // live across the whole item they had cost an s_waitcnt vmcnt(0) at the chunk loop's latch - which left registers the hardware
    // writes late at the mercy of the allocator: a spill of one, or a select on one, reads it too early.)
    const int n_lane = n0 + wn * 32 + lc;                       // couts n_lane, n_lane + 16
    const bool nok[2] = {n_lane < p.Cout, n_lane + 16 < p.Cout};

    for (int it = 0; it < n_items_blk; ++it, item += G) {
        HSIDM_STAMP(it, 0);
        float ep_add[NI][2], ep_bias[2] = {0.f, 0.f};          // [image][16-cout half]; loaded in the last chunk (see above)
#pragma unroll
        for (int q = 0; q < NI; ++q) ep_add[q][0] = ep_add[q][1] = 0.f;
        if (UP4) {                                             // the ring wraps to the next item's weights during this item
            int par_nx;
            item_tile(item + G < p.total_items ? item + G : (int)blockIdx.x, par_nx);
            w_base_nx = par_nx * p.steps_per_item;
        }
        int par = 0;
        const int it_tile = item_tile(item, par);
        const int py = par >> 1, px = par & 1;
        const int par_off = UP4 ? py * RP + px * PSTR : 0;            // this parity's 2x2 window inside the 3x3 halo
        HSIDM_SETPRIO(1);
        for (int chunk = 0; chunk < nch; ++chunk) {
            const E* hb = halo + cur * BUFE + par_off;
            // A operands: 3-deep register ring over the 4*NT sub-steps u = 4 tap + 2 q + r of the chunk (q: 32-channel slice of the tap,
            // r: 16-pixel half of every 32-pixel group), fetched two sub-steps ahead (measured: with one sub-step of lookahead every
            // sub-step waited ~300 cycles on LDS).  Sub-step u runs 2 MR MFMAs: both 16-cout halves, on the weight fragments 2q + nh.
            // (NP = 2: a sub-step carries twice the MFMAs, so ONE sub-step of lookahead covers the same LDS latency and the ring is
            // two deep - 16 registers that the low-half weight fragments need)
            constexpr int AD = C::AD;
            x8 a[AD][MR], a_lo[AP == 2 ? AD : 1][MR];
            auto a_fetch = [&](int u) __attribute__((always_inline)) {
                const int tp = u >> 2;
                const int off = (FR ? (tp >> 1) * RP + (tp & 1) * PSTR : (tp / 3) * RP + (tp % 3) * PSTR) +
                                (SPL ? (u & 1) * 32 + ((u >> 1) & 1) * RHALF : ((u >> 1) & 1) * 32 + (u & 1) * RHALF);   // SPL: u = 4 tap + 2 r + q
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    a[u % AD][mr] = *reinterpret_cast<const x8*>(hb + abase[mr] + off);
                    if constexpr (AP == 2) a_lo[u % AD][mr] = *reinterpret_cast<const x8*>(hb + C::HALO_ELEMS + abase[mr] + off);
                }
            };
            // SPL: the sparse instructions of a (q = 1) sub-step read BOTH slots of the two-deep ring, so the fragment of the next sub-step
            // is requested behind them; a (q = 0) sub-step requests its successor's up front.  (A third slot kept the lookahead but cost
            // sixteen registers the kernel does not have: the halo addresses were spilled INTO the matrix phase, each reload an
            // in-order vmcnt wait behind the weight prefetch.)
            constexpr int AL = AD - 1;
            a_fetch(0);
            if (AL == 2) a_fetch(1);
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) {
                if (!FRG && !SPL) b_issue(ring[(tap + 2) % 3]);        // weights two K steps ahead
                if (st_valid && tap == 0) {                    // staging of the next chunk, one vector per tap
                    if (st_chunk == 0) describe(st_item);
                    halo_begin(st_chunk);
                }
                if (!SPL && tap == 0 && chunk == nch - 1) {    // FiLM + bias for the epilogue (see above; SPL: loaded in the epilogue itself)
                    // the lane's channels are recomputed from the hardware lane id here: kept as loop-invariant 64-bit addresses
                    // (p.film + n, p.bias + n) they were spilled
                    int lane_s = lane_id_now();
                    asm volatile("" : "+v"(lane_s));
                    int par_;
                    const int fb0 = div_tpi(item_tile(item, par_)) * NI;
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        const int n_s = n0 + wn * 32 + 16 * nh + (lane_s & 15);
                        const int nn = n_s < p.Cout ? n_s : 0;
#pragma unroll
                        for (int q = 0; q < NI; ++q) {
                            const int fb = (fb0 + q < p.B) ? fb0 + q : fb0;
                            if (p.film) ep_add[q][nh] = p.film[(size_t)fb * p.film_stride + nn];
                        }
                        if (p.bias) ep_bias[nh] = p.bias[nn];
                    }
                }
                // Issue and commit are unconditional (past the last item they move stale but valid data into the unused
                // buffer): no branch separates them from the MFMAs, and the group barriers below ask the scheduler for
                // "2 MFMAs, a few VALU, 1 LDS read" slices instead of 8 MFMAs back to back followed by a block of VALU.
                if (!FR) {
                    if (tap < MAXHV) halo_issue_one(tap);
                } else if (tap < 2) {                          // 4 taps per chunk: vectors 0-3 at tap 0, the rest at tap 1
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (tap * 4 + i < MAXHV) halo_issue_one(tap * 4 + i);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int u = tap * 4 + kk, q = SPL ? (kk & 1) : (kk >> 1), r = SPL ? (kk >> 1) : (kk & 1);
                    if ((!SPL || q == 0) && u + AL < 4 * NT) a_fetch(u + AL);
                    if (FRG) f_issue((u + FL) % FS, (u + FL) % 4);       // weights FL fragments ahead
                    if constexpr (SPL) s_issue(((NT & 1) && tap == NT - 1) ? 1 : (tap & 1) ^ 1, kk);   // the next tap's weights (nine-tap chunks: the last tap's prefetch is the next chunk's first, moved to set 0 below)
                    if (!(HSIDM_ABL(16))) {
                        if (!FR) {                             // vector tap-3, slice kk (vector 6 of a 7-vector round: all of it at tap 8, sub-steps 2-3)
                            if (tap >= 3 && tap - 3 < MAXHV) halo_commit_part(tap - 3, cur ^ 1, kk);
                        } else if (tap >= 2 && (tap - 2) * 4 + kk < MAXHV) {   // one whole vector per sub-step, two taps after its issue
                            halo_commit_one((tap - 2) * 4 + kk, cur ^ 1);
                        }
                    }
                    if constexpr (SPL) {
                        const int set = tap & 1;
                        if (q == 0 && tap == 0 && chunk == 0) {       // first use of these accumulators: C = 0 as the inline constant (uniform branch)
                            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                                for (int nh = 0; nh < 2; ++nh) acc[mr][r][nh] = EL::mfma16(whi[SPL ? set * 4 + nh : 0], a[u % AD][mr], zero);
                        } else {
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                                for (int nh = 0; nh < 2; ++nh)
                                    acc[mr][r][nh] = EL::mfma16(whi[SPL ? set * 4 + 2 * q + nh : 0], a[u % AD][mr], acc[mr][r][nh]);
                        }
                        if (q == 1) {                              // the tap's 64 channels of half r against the sparse low halves
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr) {
                                typedef _Float16 f16x16v __attribute__((ext_vector_type(16)));
                                const f16x16v bb = __builtin_shufflevector(a[(u + AD - 1) % AD][mr], a[u % AD][mr], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                                acc[mr][r][0] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(wls[SPL ? set * 2 : 0], bb, acc[mr][r][0], wli[SPL ? set : 0], 0, 0);
                                acc[mr][r][1] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(wls[SPL ? set * 2 + 1 : 0], bb, acc[mr][r][1], wli[SPL ? set : 0], 0, 1);
                            }
                            if (u + 1 < 4 * NT) a_fetch(u + 1);
                        }
#pragma unroll
                        for (int m = 0; m < MR; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                    // 2 MFMAs (one operand, both cout halves)
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    // 1 LDS read (next sub-step's operand)
                            if (q == 1) __builtin_amdgcn_sched_group_barrier(0x002, C::XF != XF_NONE ? 3 : 1, 0);   // a slice of the staging VALU
                            else        __builtin_amdgcn_sched_group_barrier(0x002, C::XF != XF_NONE ? 6 : 2, 0);
                        }
                        if (q == 1) {
#pragma unroll
                            for (int m = 0; m < MR; ++m) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, C::XF != XF_NONE ? 3 : 1, 0);
                            }
                        }
                    } else {
                    // first sub-steps of the item: C = 0 as the MFMA's inline constant instead of 16*MR v_mov per lane and item (behind a
                    // uniform branch: a select between the constant and the accumulator would cost a v_cndmask per register)
                    if (u < 2 && chunk == 0) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
                                acc[mr][r][nh] = EL::mfma16(a[u % AD][mr], FRG ? fring[FRG ? (tap * 4 + q * 2 + nh) % FS : 0] : ring[FRG ? 0 : tap % 3][q * 2 + nh], zero);
                    } else {
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
                                acc[mr][r][nh] = EL::mfma16(a[u % AD][mr], FRG ? fring[FRG ? (tap * 4 + q * 2 + nh) % FS : 0] : ring[FRG ? 0 : tap % 3][q * 2 + nh],
                                                            acc[mr][r][nh]);
                    }
                    if (NP == 2) {                             // second pass on the weights' low halves
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
                                acc[mr][r][nh] = EL::mfma16(a[u % AD][mr], fring_lo[NP == 2 ? (tap * 4 + q * 2 + nh) % FS : 0], acc[mr][r][nh]);
                    }
                    if constexpr (AP == 2) {                   // third pass: the activations' low halves on the weights' high halves
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
                                acc[mr][r][nh] = EL::mfma16(a_lo[u % AD][mr], fring[FRG ? (tap * 4 + q * 2 + nh) % FS : 0], acc[mr][r][nh]);
                    }
#pragma unroll
                    for (int m = 0; m < MR; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                    // 2 MFMAs (one A operand, both cout halves)
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    // 1 LDS read (A operand, two sub-steps ahead)
                        __builtin_amdgcn_sched_group_barrier(0x002, C::XF != XF_NONE ? (NP == 2 ? 3 : 6) : (NP == 2 ? 1 : 2), 0);   // a slice of the staging VALU
                    }
                    if (NP == 2) {
#pragma unroll
                        for (int m = 0; m < MR * AP; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            if (AP == 2 && m < MR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // the low-half A operand
                            __builtin_amdgcn_sched_group_barrier(0x002, C::XF != XF_NONE ? 3 : 1, 0);
                        }
                    }
                    }
                    __builtin_amdgcn_sched_barrier(0);         // keep the two-sub-step LDS lookahead the source expresses
                }
                if (!FR && tap == 8 && MAXHV == 7 && !(HSIDM_ABL(16))) halo_commit_one(6, cur ^ 1);
            }
            if constexpr (SPL && (NT & 1)) {                   // nine taps per chunk: the prefetch of the last tap went to set 1, the next chunk starts on set 0
#pragma unroll
                for (int f = 0; f < 4; ++f) whi[f] = whi[SPL ? 4 + f : 0];
                wls[0] = wls[SPL ? 2 : 0];
                wls[SPL ? 1 : 0] = wls[SPL ? 3 : 0];
                wli[0] = wli[SPL ? 1 : 0];
            }
            HSIDM_STAMP(it, 10);                               // (last chunk's) MFMAs + commits issued
            lds_barrier();                                     // next halo tile complete; this one free for re-use
            if (chunk < 8) HSIDM_STAMP(it, 1 + chunk);
            cur ^= 1;
            if (st_valid) stage_advance();
        }

        // ---- epilogue of this item ------------------------------------------------------------------------------------
        HSIDM_SETPRIO(0);
        HSIDM_STAMP(it, 12);
        int b0, oy0, ox0;
        tile_coords(item, b0, oy0, ox0);
        const int it_rem = it_tile - div_tpi(it_tile) * tiles_per_img;
        const int trem = UP4 ? it_rem * 4 + par : it_rem;   // statistics slot in the image
        constexpr int US = UP4 ? 2 : 1;                                       // output pixel = US * tile pixel + parity
        const int oyb = US * oy0 + py, oxb = US * ox0 + px;                  // (py = px = 0 unless UP4)
        const int lim_h = UP4 ? p.Hin : p.Hout, lim_w = UP4 ? p.Win : p.Wout;
        float s1[NI][2], s2[NI][2];                                           // [image][16-cout half]
#pragma unroll
        for (int q = 0; q < NI; ++q) s1[q][0] = s1[q][1] = s2[q][0] = s2[q][1] = 0.f;
#pragma unroll
        for (int q = 0; q < NI; ++q)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) ep_add[q][nh] = nok[nh] ? ep_add[q][nh] + ep_bias[nh] : 0.f;
        f32x4 ep4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};          // SPL: bias + FiLM of couts 16 nh + 4 g .. + 3 (the dispatcher only
        if constexpr (SPL) {                                                  // sends whole tiles and whole cout slices this way)
            int lane_s = lane_id_now();
            asm volatile("" : "+v"(lane_s));
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                const int n4 = n0 + wn * 32 + 16 * nh + 4 * (lane_s >> 4);
                if (p.film) ep4[nh] = *reinterpret_cast<const f32x4*>(p.film + (size_t)b0 * p.film_stride + n4);
                if (p.bias) ep4[nh] += *reinterpret_cast<const f32x4*>(p.bias + n4);
            }
        }
        const bool full = SPL || (oy0 + TH <= lim_h && ox0 + TW <= lim_w && b0 + NI <= p.B && n0 + BN <= p.Cout);
        if (full) {
            // Whole tile inside the image.  2-byte stores straight from the accumulator layout cost ~200 cycles each
            // (in-kernel stamps: the epilogue took as long as 9 K steps), so the wave transposes its 64-pixel x 32-cout
            // half tiles through a private patch of the just-freed halo buffer and moves 16-byte vectors (8 couts of one
            // pixel): residual add, statistics and the store happen on those vectors.  All residual vectors of the item
            // are requested up front so that their latency hides behind the transposition.
            constexpr int LTW = (TW == 16) ? 4 : 3;
            constexpr int SCR_STR = 40;                                       // bf16 per pixel row: 32 couts + 16 B pad
            constexpr int NV = 2 * MR;                                        // 16-B vectors per lane and item
            // (fp32 mode: 4-byte patch elements, 10 KB per wave, in the free buffer's 56 KB)
            constexpr int EP = C::EP;                                         // 32-pixel row groups per pass (V2Cfg)
            S* scr = reinterpret_cast<S*>(C::OWN_PATCH ? reinterpret_cast<E*>(smem_raw + (size_t)2 * BUFE * 2 + C::POS_BYTES) : halo + (cur ^ 1) * BUFE) +
                        wave * (32 * EP * SCR_STR);
            // vector v of a pass covers pixel pl = lane/4 + 16*v: offset = (lane part, one VGPR) + (uniform part, SALU).  The lane
            // constants are rebuilt from the hardware lane id at the top of every pass: kept across the passes they were
            // spilled, and each reload was an s_waitcnt vmcnt(0) behind the previous pass's output stores.
            auto vec_base = [&](int g, int v4) __attribute__((always_inline)) -> size_t {
                const int pbase = wm * (C::BM / WM) + g * 32;
                const int img = pbase / (TH * TW);
                const int qimg = pbase - img * (TH * TW);
                const int ty = (qimg >> LTW) + v4 * (16 >> LTW);
                return (((size_t)(b0 + img) * p.Hout + oyb + US * ty) * p.Wout + oxb) * p.Cout + n0 + wn * 32;
            };
            auto run = [&](auto leaky_tag, auto res_tag) __attribute__((always_inline)) {
                constexpr bool LEAKY = decltype(leaky_tag)::value != 0;
                constexpr bool RES = decltype(res_tag)::value != 0;
                x8 rv[(RES && !F32) ? 4 : 1];                                  // residual vectors of the current pass
                f32x4 rvf[(RES && F32) ? 4 : 1][2];
                float vs1[8], vs2[8];
#pragma unroll
                for (int g = 0; g < MR; g += EP) {
                    const int nm = (MR - g) < EP ? (MR - g) : EP;              // 32-row MFMA tiles in this pass
                    const int pbase = wm * (C::BM / WM) + g * 32;
                    const int img = pbase / (TH * TW);                         // a pass never straddles two images
                    const int b = b0 + img;
                    int lane_e = lane_id_now();
                    asm volatile("" : "+v"(lane_e));
                    const int pl0 = lane_e >> 2, cq = lane_e & 3;
                    const int lc_e = lane_e & 15, lg_e = lane_e >> 4;
                    const unsigned lane_el = (unsigned)((US * (pl0 >> LTW) * p.Wout + US * (pl0 & (TW - 1))) * p.Cout + cq * 8);
                    if (RES) {                                                 // requested first: the latency hides behind the transposition
#pragma unroll
                        for (int v4 = 0; v4 < 4; ++v4)
                            if (v4 < 2 * nm) {
                                const S* rp = reinterpret_cast<const S*>(p.res) + vec_base(g, v4) + lane_el;
                                if constexpr (F32) { rvf[RES ? v4 : 0][0] = *reinterpret_cast<const f32x4*>(rp); rvf[RES ? v4 : 0][1] = *reinterpret_cast<const f32x4*>(rp + 4); }
                                else rv[RES ? v4 : 0] = *reinterpret_cast<const x8*>(rp);
                            }
                    }
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) {
                        if (m2 >= nm) break;
                        if constexpr (SPL) {
                            // transposed accumulators: the lane holds couts 16 nh + 4 lg .. + 3 of pixel lc of the half -> one 8-byte patch write
#pragma unroll
                            for (int r = 0; r < 2; ++r)
#pragma unroll
                                for (int nh = 0; nh < 2; ++nh) {
                                    float v[4];
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        v[j] = C::XF != XF_NONE ? fmaf(acc[g + m2][r][nh][j], kLn2, ep4[nh][j]) : acc[g + m2][r][nh][j] + ep4[nh][j];
                                        if (LEAKY) v[j] = v[j] > 0.f ? v[j] : 0.01f * v[j];
                                    }
                                    const x2 p0 = cvt_pair_hw<E>(v[0], v[1]), p1 = cvt_pair_hw<E>(v[2], v[3]);
                                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                    u32x2 pk;
                                    pk[0] = __builtin_bit_cast(unsigned, p0);
                                    pk[1] = __builtin_bit_cast(unsigned, p1);
                                    *reinterpret_cast<u32x2*>(reinterpret_cast<E*>(scr) + (m2 * 32 + 16 * r + lc_e) * SCR_STR + 16 * nh + 4 * lg_e) = pk;
                                }
                            continue;
                        }
                        // patch row = 32 m2 + 16 (half r) + pixel of the half; the lane holds pixels 4 lg .. + 3 of couts 16 nh + lc
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                                for (int j = 0; j < 4; j += 2) {                // pixels j, j + 1: one packed conversion
                                    float v[2];
#pragma unroll
                                    for (int e = 0; e < 2; ++e) {
                                        const float av = acc[g + m2][r][nh][j + e], ea = ep_add[NI == 1 ? 0 : img][nh];
                                        v[e] = (C::XF != XF_NONE && !F32) ? fmaf(av, kLn2, ea) : av + ea;
                                        if (LEAKY) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
                                        // no residual: statistics from the fp32 values in the accumulator layout (the lane owns two couts: 2 VALU
                                        // per value, two exchanges between the lane quarters) instead of unpacking the stored vectors and the
                                        // 15-move butterfly below; the rounding noise of the store is zero-mean and 2^-9 relative
                                        if (!RES) { s1[NI == 1 ? 0 : img][nh] += v[e]; s2[NI == 1 ? 0 : img][nh] = fmaf(v[e], v[e], s2[NI == 1 ? 0 : img][nh]); }
                                    }
                                    const int at = (m2 * 32 + 16 * r + 4 * lg_e + j) * SCR_STR + 16 * nh + lc_e;
                                    if constexpr (F32) {
                                        scr[at] = v[0];
                                        scr[at + SCR_STR] = v[1];
                                    } else {
                                        const x2 pr = cvt_pair_hw<E>(v[0], v[1]);
                                        reinterpret_cast<E*>(scr)[at] = pr[0];
                                        reinterpret_cast<E*>(scr)[at + SCR_STR] = pr[1];
                                    }
                                }
                    }
                    HSIDM_STAMP(it, 9);
                    if (NI == 2 ? g % 2 == 0 : g == 0) {                       // (NI = 2: an image is two row groups)
#pragma unroll
                        for (int k = 0; k < 8; ++k) vs1[k] = vs2[k] = 0.f;
                    }
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
                            S* op = reinterpret_cast<S*>(p.out) + vec_base(g, v4) + lane_el;
                            if (!HSIDM_ABL(1)) { *reinterpret_cast<f32x4*>(op) = o0; *reinterpret_cast<f32x4*>(op + 4) = o1; }
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
                        if (!HSIDM_ABL(1)) *reinterpret_cast<x8*>(reinterpret_cast<E*>(p.out) + vec_base(g, v4) + lane_el) = o;
                        }
                        if (RES || SPL) {                                      // (SPL: a lane of the accumulator layout holds eight couts)
#pragma unroll
                            for (int k = 0; k < 8; ++k) { vs1[k] += f[k]; vs2[k] = fmaf(f[k], f[k], vs2[k]); }
                        }
                    }
                    HSIDM_STAMP(it, 11);
                    if ((RES || SPL) && p.stats && (NI == 2 ? (g + EP) % 2 == 0 : g + EP >= MR)) {
                        // Lanes with equal (lane & 3) hold the same 8 couts: fold the 16 of them together with a halving
                        // butterfly -- at every level a lane hands the half of its values its partner keeps and receives
                        // the half it keeps itself: 8+4+2+1 = 15 cross-lane moves instead of 4*16, and every lane ends
                        // up with ONE finished entry, idx = 8*bit2 + 4*bit3 + 2*bit4 + bit5 (idx < 8: sum of cout idx,
                        // else sum of squares of cout idx - 8).
                        const bool hi0 = (lane_e & 4) != 0, hi1 = (lane_e & 8) != 0, hi2 = (lane_e & 16) != 0, hi3 = (lane_e & 32) != 0;
                        float a8[8], a4[4], a2[2];
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            a8[i] = (hi0 ? vs2[i] : vs1[i]) + lane_xor<4>(hi0 ? vs1[i] : vs2[i], lane_e);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            a4[i] = (hi1 ? a8[i + 4] : a8[i]) + lane_xor<8>(hi1 ? a8[i] : a8[i + 4], lane_e);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            a2[i] = (hi2 ? a4[i + 2] : a4[i]) + lane_xor<16>(hi2 ? a4[i] : a4[i + 2], lane_e);
                        const float a1 = (hi3 ? a2[1] : a2[0]) + lane_xor<32>(hi3 ? a2[0] : a2[1], lane_e);
                        int sub;
                        if (NI == 1) sub = wm;
                        else if (WM == 1) sub = 0;
                        else sub = wm % (WM / 2 > 0 ? WM / 2 : 1);
                        const int idx = ((lane_e >> 2) & 1) * 8 + ((lane_e >> 3) & 1) * 4 + ((lane_e >> 4) & 1) * 2 + (lane_e >> 5);
                        float* dst = reinterpret_cast<float*>(p.stats + ((size_t)b * (tiles_per_img * (US * US) * C::SUBS) + trem * C::SUBS + sub) * p.Cout + n0 + wn * 32);
                        dst[(cq * 8 + (idx & 7)) * 2 + (idx >> 3)] = a1;
                    }
                }
            };
            if (p.act == ACT_LEAKY) { if (p.res) run(SlotTag<1>{}, SlotTag<1>{}); else run(SlotTag<1>{}, SlotTag<0>{}); }
            else                    { if (p.res) run(SlotTag<0>{}, SlotTag<1>{}); else run(SlotTag<0>{}, SlotTag<0>{}); }
            HSIDM_STAMP(it, 14);
            if (!C::OWN_PATCH) lds_barrier();                                // the patch is part of the next staging buffer (else: a region of its own)
        } else {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int pbase = wm * (C::BM / WM) + mr * 32;
            const int img = pbase / (TH * TW);                       // a 32-pixel group never straddles images
            const int b = b0 + img;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = 16 * r + 4 * lg + j;
                        const int q = pbase + row - img * (TH * TW);
                        const int ty = q / TW, tx = q - ty * TW;
                        const int oy = oyb + US * ty, ox = oxb + US * tx;
                        const float ea = ep_add[NI == 1 ? 0 : img][nh];
                        float v = (C::XF != XF_NONE && !F32) ? fmaf(acc[mr][r][nh][j], kLn2, ea) : acc[mr][r][nh][j] + ea;
                        if (!(nok[nh] && b < p.B && oy0 + ty < lim_h && ox0 + tx < lim_w)) continue;
                        if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
                        const size_t o = (((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + n_lane + 16 * nh;
                        if (p.res) v = p.res_scale * v + (float)reinterpret_cast<const S*>(p.res)[o];
                        const S st = F32 ? (S)v : (S)v;
                        if (!(HSIDM_ABL(1))) reinterpret_cast<S*>(p.out)[o] = st;
                        const float sv = (float)st;
                        s1[NI == 1 ? 0 : img][nh] += sv;
                        s2[NI == 1 ? 0 : img][nh] += sv * sv;
                    }
        }
        }
        if (!SPL && p.stats && (!full || !p.res)) {
            // wave partial over its pixels: combine the four lane quarters (lg), lanes 0..15 write one entry per cout half
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                int lane_w = lane_id_now();                               // rebuilt here, not kept across the item (see lane_xor)
                asm volatile("" : "+v"(lane_w));
                int img, sub;
                if (NI == 1) { img = 0; sub = wm; }
                else if (WM == 1) { img = q; sub = 0; }
                else { img = (wm * (C::BM / WM)) / (TH * TW); sub = wm % (WM / 2 > 0 ? WM / 2 : 1); if (q != img) continue; }
                const int b = b0 + img;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    float a = s1[q][nh] + lane_xor<16>(s1[q][nh], lane_w);
                    float d = s2[q][nh] + lane_xor<16>(s2[q][nh], lane_w);
                    a += lane_xor<32>(a, lane_w);
                    d += lane_xor<32>(d, lane_w);
                    if (lg == 0 && nok[nh] && b < p.B)
                        p.stats[((size_t)b * (tiles_per_img * (US * US) * C::SUBS) + trem * C::SUBS + sub) * p.Cout + n_lane + 16 * nh] = make_float2(a, d);
                }
            }
        }
        HSIDM_STAMP(it, 13);
    }
}

}  // namespace hsidm
