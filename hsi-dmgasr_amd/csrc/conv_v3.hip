// conv_v3: 3x3 stride-1 convolution with GroupNorm + SiLU prologue for the 64-cout layers of the 128x128 level
// (ResnetBlock convs at inner_channel, reference unet.py:83-111), bf16.
//
// Why another kernel: on conv_v2 these layers are instruction-issue bound, not MFMA bound (SQ counters, batch 120,
// 64->64 at 128x128: 13 VALU + 5.6 SALU + 1.6 LDS instructions per MFMA, MFMA pipe 24 % busy, 254 us against an HBM
// floor of 80 us): a 128-pixel x 64-cout item has only 72 MFMAs per wave to carry its halo description, GroupNorm
// transform, epilogue and bookkeeping.  conv_v3 doubles the pixels per item at the same cout slice:
//   * tile = 16x16 pixels x 64 couts, 4 waves as 2 (pixel halves of 8 rows) x 2 (cout halves): 4 MFMA tiles per wave and
//     k-slice instead of 2, so every A-fragment read, weight fragment, wait and loop instruction is amortised over twice
//     the MFMAs, the 18x18 halo is 1.27x the tile instead of 1.41x, and per-item costs are paid once per 256 pixels;
//   * the halo tile is SINGLE-buffered (50 KB; double-buffered it would not leave room for two workgroups per CU):
//     commit (transform) -> barrier -> 36 k-slices of MFMAs -> barrier.  The raw vectors of the next chunk are requested
//     before the MFMA phase and held in registers through it; the overlap of one workgroup's transform/epilogue with
//     MFMAs comes from the OTHER workgroup on the CU;
//   * weights stream through conv_v2's 3-step register ring, epilogue / statistics / FiLM handling are conv_v2's.
// Matrix instruction: v_mfma_f32_16x16x32 (round 3).  Same cycles per FLOP as the 32x32x16 form, but the chip holds a higher clock
// on it under its power management (tools/ubench/mfma_shape.hip on this pool: 1 310 vs 1 150 TFLOP/s in a loop shaped like this
// one, +14 %; MI355X_MICROARCH.md "DVFS give-back" item 7).  A 16-row A operand is ONE tile row of 16 pixels: lane (c = lane % 16,
// g = lane / 16) reads channels 8g .. 8g+7 of the 32-channel slice of pixel column c; with 160-byte pixels the four 16-lane service
// groups of a ds_read_b128 cover all 64 banks once at every tap (144-byte pixels would be 2-way).  The B operand (16 couts x 32
// channels) is read from the SAME packed weight order as before - its 16-byte units are only addressed per lane differently.
// Requirements (api.hip falls back to conv_v2 otherwise): Cout == 64, H % 16 == 0, W % 16 == 0, no upsampling.
#include "conv_v2.h"
#include <cstdlib>
#include "../../include/hsidm.h"

namespace hsidm {

namespace v3 {
constexpr int TH = 16, TW = 16, BM = 256, BN = 64, BK = 64;
constexpr int HROWS = TH + 2, HCOLS = TW + 2, HPIX = HROWS * HCOLS;      // 18 x 18
constexpr int PSTR = BK + 16, VPP = BK / 8;                                // 160-byte pixels (see above); the last 32 B are padding
constexpr int RP = HCOLS * PSTR;                                           // halo row pitch
constexpr int HVEC = HPIX * VPP, MAXHV = (HVEC + 255) / 256;               // 2592 vectors, 11 per thread
constexpr int HALO_ELEMS = HROWS * RP;
constexpr int SCR_STR = 40;                                                // bf16 per row of an epilogue patch: 32 couts + 16 B pad
// halo tile + halo position table + one 32-pixel x 32-cout transposition patch per wave.  The patches have a region of their own
// (they used to sit in the halo tile): a wave leaves its epilogue for the next item's commit without a workgroup barrier
constexpr size_t PATCH_OFF = (size_t)HALO_ELEMS * 2 + (size_t)MAXHV * 256 * 4;
constexpr size_t LDS_BYTES = PATCH_OFF + (size_t)4 * 32 * SCR_STR * 2;
static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
}  // namespace v3

// WN_ = 2: the 64-cout layers (waves = 2 pixel halves x 2 cout halves, 4 MFMA tiles per wave).
// WN_ = 1, NCHW_: the UNet's final Block (GroupNorm + SiLU + conv 64 -> 3, reference unet.py:231,262): one 32-cout slice of which
//   3 couts exist, waves = 4 pixel quarters, 2 MFMA tiles per wave, fp32 NCHW output straight from the accumulators (no FiLM,
//   residual or statistics) - the last launch of a bf16 step that ran on the generic kernel (415 us at batch 240).
// E: element type (bf16 | fp16); NP = 2: a second MFMA pass on the weights' low halves (conv_v2.h, V2Cfg) - these layers run the
// matrix pipe a third of the time, so the pass is nearly free and the weights of the level whose output IS the network's output
// carry ~18-19 significant bits (subnormal low halves, 6e-8 granularity); the weight ring is then 6 (high, low) fragment pairs fetched five k-slices ahead (the same registers).
// SPL (fp16, NP = 2, WN = 2): the second weight pass on v_smfmac_f32_16x16x64_f16 with the low halves 2:4-compressed (hsidm.h: w_v2_ls /
// w_v2_li).  The sparse operand is the instruction's A side, so in this form the WEIGHTS are the A operand of both passes and the
// accumulators come out transposed: lane (c = pixel column, g) holds couts 16 nh + 4g .. + 3 of its pixel.  One sparse instruction
// covers the 64 channels of a tap (both 32-channel slices: its dense operand is simply the two activation fragments of the tap side by
// side) at ~1.2x the cycles of one dense 16x16x32 (tools/ubench/smfmac_rate.hip) - the second pass costs ~0.6 of its dense form, and
// the low halves lose the smaller two of every four values: 20 % of their energy, i.e. of a 2^-12 correction (tests/precision_emul.py,
// w=x2s: the 20-step chain moves from 5.38e-4 to 5.50e-4).  Sub-steps run in the order (tap, row r, slice q): the tap's four dense
// weight fragments stay in registers for both rows; two register sets alternate between taps, one tap of lookahead.
// PROJ (one-pass forms: NP = 1, WN = 2): the launch also carries a ResnetBlock's 1x1 residual projection (hsidm_conv_desc.ph[1]; reference
// unet.py:102-103,110) as up to three more 64-channel chunks of ONE tap accumulated into the same tile - see "PROJ" at issue_all /
// commit_all and the three static slots behind the 9-tap loop.  (Round 4 built it for the sparse-lo form - the kernel set the benchmark ran then;
// profiles/r04_final/ab_fused_proj.txt, stamps_v3_proj.txt - round 6 for the one-pass sets and retired the sparse combination.)
typedef _Float16 f16x16v __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int WN_, bool NCHW_, typename E = bf16, int NP = 1, bool SPL = false, bool PROJ = false>
__global__ __launch_bounds__(256, 2) void conv_v3_kernel(const ConvV2Params p) {
    using namespace v3;
    using EL = Elem<E>;
    using x8 = typename EL::x8;
    using x2 = typename EL::x2;
    static_assert(!SPL || (NP == 2 && WN_ == 2 && !NCHW_ && !__is_same(E, bf16)), "sparse low halves: fp16, two passes, the 64-cout form");
    static_assert(!PROJ || (!SPL && NP == 1 && WN_ == 2 && !NCHW_), "fused 1x1 projection: the one-pass 64-cout forms");
    constexpr bool FRG = NP == 2 && !SPL;
    constexpr int FS = 6, FL = 4;                               // a k-slice pair keeps its two fragments for both row halves: FL <= FS - 2
    constexpr int WN = WN_, WM = 4 / WN_, MR = 8 / WM;          // a wave owns 16 / WM tile rows = MR MFMA tiles of two rows
    // NCHW form (Cout <= 16, checked by the dispatcher: the UNet's final 64 -> 3 conv): only the first 16-cout half of the padded
    // 32-cout slice holds weights - the second half's MFMAs would multiply zeros (half of this launch's matrix instructions)
    constexpr int NHN = NCHW_ ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* halo = reinterpret_cast<E*>(smem_raw);
    if constexpr (!__is_same(E, bf16)) fp16_saturating_stores();     // (common.h)
    int* pos_tab = reinterpret_cast<int*>(smem_raw + (size_t)HALO_ELEMS * 2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lc = lane & 15, lg = lane >> 4;                  // MFMA 16x16x32: row / column lc, k group lg
    const int G = gridDim.x;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    // tile index -> (image, tile row, tile column): shifts when the tile grid is a power of two (the UNet's maps are); a division by
    // a run-time value is ~20 scalar instructions, and an item needs eight of them
    auto div_tpi = [&](int x) __attribute__((always_inline)) -> int { return p.tpi_shift >= 0 ? x >> p.tpi_shift : x / tiles_per_img; };
    auto div_tx = [&](int x) __attribute__((always_inline)) -> int { return p.tx_shift >= 0 ? x >> p.tx_shift : x / p.tiles_x; };
    const int ctot = p.C0 + p.C1;
    const int nch = p.nchunks;
    // PROJ: p.pchunks more chunks of ONE tap each - the 1x1 projection of a second input (ResnetBlock.res_conv, reference unet.py:102-103,110)
    // accumulated into the same tile: staged raw (no GroupNorm / SiLU), multiplied at the centre tap by weight steps 9 nch .. (the host
    // scales them by log2(e): the epilogue undoes the factor the SiLU staging carries on EVERY product)
    const int nct = nch + (PROJ ? p.pchunks : 0);

    int item = blockIdx.x;                                      // one cout slice: item = pixel tile
    const int n_items_blk = (p.total_items - item + G - 1) / G;

    // ---- weight stream (conv_v2's order: [step = chunk*9 + tap][cout/32][kk][lane][8]) ----------------------------------
    const int nsw = p.Cout_pad >> 5;
    // B operand (n, q) of a 32-cout slice: cout 16n + lc, channels 32q + 8 lg .. +7 = the 16-byte unit at fragment 2q + lg/2,
    // lane 32 (lg % 2) + 16n + lc of the packed order; fragment index e = 2q + n below
    const size_t wl_off = ((size_t)wn * 4 * 64 + (lg >> 1) * 64 + 32 * (lg & 1) + lc) * 8;
    const E* wlane = reinterpret_cast<const E*>(p.w) + wl_off;
    const E* wlane_lo = reinterpret_cast<const E*>(NP == 2 ? p.w_lo : p.w) + wl_off;
    auto frag_off = [](int e) __attribute__((always_inline)) -> int { return (e >> 1) * (2 * 64 * 8) + (e & 1) * (16 * 8); };
    const size_t wstep_stride = (size_t)nsw * 4 * 64 * 8;
    x8 ring[(FRG || SPL) ? 1 : 3][4];
    x8 fring[FRG ? FS : 1], fring_lo[FRG ? FS : 1];
    // SPL: register sets (tap & 1): four dense fragments e = 2q + nh, the two sparse halves, one index word (low / high 16 bits = cout half)
    x8 whi[SPL ? 8 : 1], wls[SPL ? 4 : 1];
    int wli[SPL ? 2 : 1];
    const E* lsp = reinterpret_cast<const E*>(SPL ? p.w_ls : p.w) + ((size_t)wn * 2 * 64 + lane) * 8;
    const int* lip = (SPL ? p.w_li : reinterpret_cast<const int*>(p.w)) + (size_t)wn * 64 + lane;
    int wnext = 0;
    auto s_issue = [&](int set, int part) __attribute__((always_inline)) {      // part 0..3 of weight step wnext into register set `set`
        whi[SPL ? set * 4 + part : 0] = *reinterpret_cast<const x8*>(wlane + (size_t)wnext * wstep_stride + frag_off(part));
        if (part < 2) wls[SPL ? set * 2 + part : 0] = *reinterpret_cast<const x8*>(lsp + ((size_t)wnext * nsw * 2 + part) * (64 * 8));
        if (part == 2) wli[SPL ? set : 0] = lip[(size_t)wnext * nsw * 64];
        if (part == 3) wnext = (wnext + 1 == p.steps_per_item) ? 0 : wnext + 1;
    };
    auto b_issue = [&](x8 (&dst)[4]) __attribute__((always_inline)) {
        const E* src = wlane + (size_t)wnext * wstep_stride;
        if (!HSIDM_ABL(8)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dst[kk] = *reinterpret_cast<const x8*>(src + frag_off(kk));
        }
        wnext = (wnext + 1 == p.steps_per_item) ? 0 : wnext + 1;
    };
    auto f_issue = [&](int slot, int kk) __attribute__((always_inline)) {
        const size_t off = (size_t)wnext * wstep_stride + frag_off(kk);
        fring[FRG ? slot : 0] = *reinterpret_cast<const x8*>(wlane + off);
        fring_lo[FRG ? slot : 0] = *reinterpret_cast<const x8*>(wlane_lo + off);
        if (kk == 3) wnext = (wnext + 1 == p.steps_per_item) ? 0 : wnext + 1;
    };

    // ---- halo staging: vector i of this thread = halo pixel (tid/8 + 32 i), channels 8*(tid%8).. of the chunk ------------
    const int cv = tid & 7;
#pragma unroll
    for (int i = 0; i < MAXHV; ++i) {
        const int hp = (tid >> 3) + i * 32;
        const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
        pos_tab[i * 256 + tid] = hp < HPIX ? (((hy * RP + hx * PSTR + cv * 8) << 10) | (hy << 5) | hx) : -1;   // LDS offset | hy | hx
    }
    int hv_pix[MAXHV];
    u32x4 hreg[MAXHV];
    unsigned abh[8];
    int st_item = item, st_chunk = 0, st_b = 0;
    bool st_valid = true, st_cok = true;
    // work item -> pixel tile.  Blocks b, b+8, b+16, ... share an XCD (and its L2); with the identity map they would hold tiles
    // 8 apart while the tiles in between - the ones that share their halo rows and columns - sit on the other seven XCDs.
    // xcd_m = 8 when the tile count allows: XCD x walks the contiguous range [x*M/8, (x+1)*M/8), 64 neighbouring tiles at a time.
    auto tile_of = [&](int it) __attribute__((always_inline)) -> int {
        return p.xcd_m > 0 ? (it & 7) * (p.m_tiles >> 3) + (it >> 3) : it;     // xcd_m is 0 or 8
    };
    auto describe = [&](int it_) __attribute__((always_inline)) {
        if (HSIDM_ABL(32) && it_ != (int)blockIdx.x) return;     // (diagnostic builds: the first item's addresses for every item)
        const int it = tile_of(it_);
        const int b = div_tpi(it);
        const int tr = it - b * tiles_per_img;
        const int try_ = div_tx(tr);
        const int oy0 = try_ * TH, ox0 = (tr - try_ * p.tiles_x) * TW;
        st_b = b;
        int posv[MAXHV];                                        // all table reads first: one LDS round trip instead of eleven
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) posv[i] = pos_tab[i * 256 + tid];
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) {
            const int pos = posv[i];
            const int iy = oy0 + ((pos >> 5) & 31) - 1, ix = ox0 + (pos & 31) - 1;
            const bool ok = pos >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            hv_pix[i] = ok ? (b * p.Hin + iy) * p.Win + ix : -1;
        }
    };
    // A chunk's staging request: issue_begin (channel slice, source pointer, the GroupNorm pairs) and issue_one(i), the load of halo
    // vector i - eleven back to back ahead of the matrix phase.  (Round 6 measured two other placements, neither kept: spread over the
    // phase's taps, conv_v2's schedule - neutral to 1 % slower, profiles/r06_ab/ab_spread_issue.txt; behind the weight request of tap 6, so
    // that none of the phase's in-order weight waits depends on them - plain forms 1 % faster, the projection form spills,
    // profiles/r06_ab/ab_late_issue.txt.)
    const E* is_src = reinterpret_cast<const E*>(p.src0);
    int is_cs = 0;
    auto issue_begin = [&](int chunk) __attribute__((always_inline)) {
        const int c = chunk * BK + cv * 8;
        st_cok = c < ctot;
        const int cc = st_cok ? c : 0;
        if (cc < p.C0) { is_src = reinterpret_cast<const E*>(p.src0) + cc; is_cs = p.C0; }
        else           { is_src = reinterpret_cast<const E*>(p.src1) + (cc - p.C0); is_cs = p.C1; }
        {   // log2(e)-scaled fp16x2 part of the GroupNorm table (conv_v2.h: gn_params, silu_log2e)
            const u32x4* t = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned*>(p.gn_ab) + (size_t)3 * p.B * ctot + (size_t)st_b * ctot + cc);
            const u32x4 lo = t[0], hi = t[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) { abh[k] = lo[k]; abh[4 + k] = hi[k]; }
        }
    };
    auto issue_one = [&](int i) __attribute__((always_inline)) {
        if (HSIDM_ABL(4)) return;
        const int pix = hv_pix[i] >= 0 ? hv_pix[i] : 0;
        hreg[i] = *reinterpret_cast<const u32x4*>(is_src + (size_t)pix * is_cs);
    };
    auto issue_all = [&](int chunk) __attribute__((always_inline)) {
        if (PROJ && chunk >= nch) {                             // workgroup-uniform
            // projection chunk: only the tile's own 16 x 16 pixels, 2048 vectors = eight per thread: vector i = pixel (row 2 i + tid / 128,
            // column (tid / 8) % 16), channels 8 (tid % 8) .. of the chunk - addresses a row pair apart, no halo table, no bounds
            const int c = (chunk - nch) * BK + cv * 8;
            st_cok = c < p.PC0 + p.PC1;
            const int cc = st_cok ? c : 0;
            const E* src;
            int cs;
            if (cc < p.PC0) { src = reinterpret_cast<const E*>(p.psrc0) + cc; cs = p.PC0; }
            else            { src = reinterpret_cast<const E*>(p.psrc1) + (cc - p.PC0); cs = p.PC1; }
            const int it = tile_of(st_item);
            const int b = div_tpi(it);
            const int tr = it - b * tiles_per_img;
            const int try_ = div_tx(tr);
            const int oy0 = try_ * TH, ox0 = (tr - try_ * p.tiles_x) * TW;
            const E* base = src + (size_t)((b * p.Hin + oy0 + (tid >> 7)) * p.Win + ox0 + ((tid >> 3) & 15)) * cs;
            const size_t pair = (size_t)2 * p.Win * cs;
#pragma unroll
            for (int i = 0; i < 8; ++i) hreg[i] = *reinterpret_cast<const u32x4*>(base + i * pair);
            return;
        }
        issue_begin(chunk);
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) issue_one(i);
    };
    // dead slots of the last (partial) vector round store into the row padding instead of being branched around, so the
    // eleven transforms form one basic block the scheduler can interleave (conv_v2.h: dead_off)
    const int dead_off = (tid / HCOLS) * RP + (tid % HCOLS) * PSTR + BK;      // (the padding of halo pixel `tid`)
    auto commit_all = [&](bool pj) __attribute__((always_inline)) {
        if (PROJ && pj) {
            // projection chunk: the tile's own pixels, raw, stored ONE halo row and column further in than their place in the halo tile
            // (the matrix phase then reads them with the offsets of tap 8); vector i of issue_all -> rows 2 i apart: immediate offsets
            E* dst = halo + ((tid >> 7) + 2) * RP + (((tid >> 3) & 15) + 2) * PSTR + cv * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u32x4 ou = hreg[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) ou[k] = st_cok ? ou[k] : 0u;
                *reinterpret_cast<u32x4*>(dst + i * 2 * RP) = ou;
            }
            return;
        }
        if (HSIDM_ABL(16)) return;
        int posv[MAXHV];                                        // table reads before the first halo store (they may alias for the compiler)
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) posv[i] = pos_tab[i * 256 + tid];
        if (HSIDM_ABL(2)) {                                     // (diagnostic builds: the raw vectors, no GroupNorm + SiLU)
#pragma unroll
            for (int i = 0; i < MAXHV; ++i) {
                u32x4 ou = hreg[i];
                const bool live = st_cok && hv_pix[i] >= 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) ou[k] = live ? ou[k] : 0u;
                *reinterpret_cast<u32x4*>(halo + (posv[i] >= 0 ? (posv[i] >> 10) : dead_off)) = ou;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < MAXHV; ++i) {
            const int pos = posv[i];
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] = EL::lo(hreg[i][k]);
                v[2 * k + 1] = EL::hi(hreg[i][k]);
            }
            // the transform in STAGES over the vector's eight elements (scheduling barriers in between): element by element the
            // compiler emits one dependent chain fma -> exp -> add -> rcp -> mul per element, with an s_nop behind every
            // transcendental and nothing to issue while its result is in flight (183 s_nop in the 872 instructions of this block; staged: 16
            // in 697; the hi + lo launches -0.7 ... -1.4 %, the rest unchanged: the partner workgroup's MFMA phase was filling most of those slots)
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], h2_lo(abh[k]), h2_hi(abh[k]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = __builtin_amdgcn_exp2f(-v[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = 1.0f + e[k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = __builtin_amdgcn_rcpf(e[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] * e[k];
            __builtin_amdgcn_sched_barrier(0);
            x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (E)v[k];
            u32x4 ou = __builtin_bit_cast(u32x4, o);
            const bool live = st_cok && hv_pix[i] >= 0;         // zero padding stays zero (pad AFTER activation)
#pragma unroll
            for (int k = 0; k < 4; ++k) ou[k] = live ? ou[k] : 0u;
            *reinterpret_cast<u32x4*>(halo + (pos >= 0 ? (pos >> 10) : dead_off)) = ou;
        }
    };

    // ---- MFMA fragment bases: pair mr = tile rows (16/WM) wm + 2 mr, +1 (one 16-row A operand each); lane = (column lc, k group lg) ----
    int abase[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) abase[mr] = (wm * (16 / WM) + mr * 2) * RP + lc * PSTR + 8 * lg;
    f32x4 acc[MR][2][2];                                        // [tile-row pair][row of the pair][16-cout half]


    // prologue: two weight steps in flight, raw vectors of (first item, chunk 0) requested
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
    issue_all(0);
    auto stage_advance = [&]() __attribute__((always_inline)) {
        if (st_chunk + 1 < nct) st_chunk += 1;
        else { st_chunk = 0; st_item += G; }
        st_valid = st_item < p.total_items;
    };

    for (int it = 0; it < n_items_blk; ++it, item += G) {
        HSIDM_STAMP(it, 0);
        const int tile = tile_of(item);
        const int b = div_tpi(tile);
        float ep_add[2] = {0.f, 0.f}, ep_bias[2] = {0.f, 0.f};  // FiLM / bias of the lane's couts [16-cout half]: loaded in the last chunk
        f32x4 ep4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // SPL: bias + FiLM of couts 16 nh + 4g .. + 3, loaded in the epilogue

        // (PROJ: the 9-tap loop covers the 3x3 chunks, the projection's chunks follow as three static slots below)
        const int n_loop = PROJ ? nch : nct;
        for (int chunk = 0; chunk < n_loop; ++chunk) {
            commit_all(chunk >= nch);                           // hreg holds (item, chunk): transform -> LDS
            if (chunk == 0) HSIDM_STAMP(it, 1);
            stage_advance();
            if (st_valid) {                                     // request the next chunk's raw vectors; they fly during the MFMAs
                if (st_chunk == 0) describe(st_item);
                issue_all(st_chunk);
            }
            if (chunk == 0) HSIDM_STAMP(it, 2);
            lds_barrier();
            if (chunk == 0) HSIDM_STAMP(it, 3);
            if (chunk == nch - 1) {
                // FiLM + bias of the lane's two couts, needed by the epilogue: ordinary (compiler-tracked) loads issued HERE, ahead of
                // the last chunk's MFMA phase.  vmcnt retires in order and the phase issues 36+ weight loads behind them, so the
                // counted waits on those weights already cover them: the ISA shows no additional s_waitcnt for these values.
                // (Rounds 1-3 fetched them at the start of the item with loads the compiler did not track - kept as a tracked value
                // across the whole item they had cost an s_waitcnt vmcnt(0) per item - which left a register the hardware writes
                // late at the mercy of the allocator: a spill or a select on it reads it too early.  In-box A/B: equal time.)
                int lane_s = lane_id_now();
                asm volatile("" : "+v"(lane_s));
                if constexpr (SPL) {
                    // (the eight per-cout values of this form are loaded at the start of the epilogue instead: sixteen registers held
                    // through the last chunk's matrix phase made the allocator spill inside it)
                } else {
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const int n_raw = wn * 32 + 16 * nh + (lane_s & 15);
                    const int n_s = (!NCHW_ || n_raw < p.Cout) ? n_raw : 0;
                    if (p.film) ep_add[nh] = p.film[(size_t)b * p.film_stride + n_s];
                    if (p.bias) ep_bias[nh] = p.bias[n_s];
                }
                }
            }
            HSIDM_SETPRIO(1);
            // A operands: ring over the 36 sub-steps w = 4 tap + 2 q + r of the chunk (q: 32-channel slice of the tap, r: row of the
            // tile-row pair); sub-step w runs 2 MR (x NP) MFMAs - both 16-cout halves on the slice's two weight fragments 4 tap + 2 q + n
            if constexpr (SPL) {
                // sub-step w = 4 tap + 2 r + q; activations: two-deep ring.  The sparse instructions of a (q = 1) sub-step read BOTH slots,
                // so the fragment of the next sub-step is requested behind them; a (q = 0) sub-step requests its successor's up front
                // (conv_v2.h, SPL: a third slot costs sixteen registers and the allocator then spills around the chunk barrier)
                x8 a[2][MR];
                auto a_fetch = [&](int w) __attribute__((always_inline)) {
                    const int tp = w >> 2;
                    const int off = (tp / 3 + ((w >> 1) & 1)) * RP + (tp % 3) * PSTR + (w & 1) * 32;
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr) a[w % 2][mr] = *reinterpret_cast<const x8*>(halo + abase[mr] + off);
                };
                a_fetch(0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int set = tap & 1;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int w = tap * 4 + kk, r = kk >> 1, q = kk & 1;
                        if (q == 0) a_fetch(w + 1);
                        s_issue(tap == 8 ? 1 : set ^ 1, kk);       // the next tap's weights (tap 8: the next chunk's first tap, moved to set 0 below)
                        if (q == 0 && tap == 0 && chunk == 0) {   // first use of these accumulators: C = 0 as the inline constant (uniform branch)
                            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                                for (int mr = 0; mr < MR; ++mr) acc[mr][r][nh] = EL::mfma16(whi[SPL ? set * 4 + nh : 0], a[w % 2][mr], zero);
                        } else {
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                                for (int mr = 0; mr < MR; ++mr)
                                    acc[mr][r][nh] = EL::mfma16(whi[SPL ? set * 4 + 2 * q + nh : 0], a[w % 2][mr], acc[mr][r][nh]);
                        }
                        if (q == 1) {                             // the tap's 64 channels of row r against the sparse low halves
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                                for (int mr = 0; mr < MR; ++mr) {
                                    const f16x16v bb = __builtin_shufflevector(a[(w + 1) % 2][mr], a[w % 2][mr], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                                    if (nh == 0) acc[mr][r][0] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(wls[SPL ? set * 2 : 0], bb, acc[mr][r][0], wli[SPL ? set : 0], 0, 0);
                                    else         acc[mr][r][1] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(wls[SPL ? set * 2 + 1 : 0], bb, acc[mr][r][1], wli[SPL ? set : 0], 0, 1);
                                }
                            if (w + 1 < 36) a_fetch(w + 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // nine taps per chunk: the prefetch of tap 8 went to set 1, where tap 0 of the next chunk expects set 0
#pragma unroll
                for (int f = 0; f < 4; ++f) whi[f] = whi[SPL ? 4 + f : 0];
                wls[0] = wls[SPL ? 2 : 0];
                wls[SPL ? 1 : 0] = wls[SPL ? 3 : 0];
                wli[0] = wli[SPL ? 1 : 0];
            } else {
            constexpr int AD = NP == 2 ? 2 : 3;                 // (conv_v2.h: one sub-step of lookahead when it carries twice the MFMAs)
            x8 a[AD][MR];
            auto a_fetch = [&](int w) __attribute__((always_inline)) {
                const int tp = w >> 2;
                const int off = (tp / 3 + (w & 1)) * RP + (tp % 3) * PSTR + ((w >> 1) & 1) * 32;
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) a[w % AD][mr] = *reinterpret_cast<const x8*>(halo + abase[mr] + off);
            };
            a_fetch(0);
            if (AD == 3) a_fetch(1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (!FRG) b_issue(ring[(tap + 2) % 3]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int w = tap * 4 + kk, q = kk >> 1, r = kk & 1;
                    if (w + AD - 1 < 36) a_fetch(w + AD - 1);
                    if (FRG) f_issue((w + FL) % FS, (w + FL) % 4);
                    if (HSIDM_ABL(128)) {                        // (diagnostic builds: no matrix instructions)
                    } else if (w < 2 && chunk == 0) {           // first use of these accumulators: C = 0 as the MFMA's inline constant (uniform branch)
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int nh = 0; nh < NHN; ++nh)
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr)
                                acc[mr][r][nh] = EL::mfma16(a[w % AD][mr], FRG ? fring[FRG ? (tap * 4 + q * 2 + nh) % FS : 0] : ring[FRG ? 0 : tap % 3][q * 2 + nh], zero);
                    } else {
#pragma unroll
                        for (int nh = 0; nh < NHN; ++nh)
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr)
                                acc[mr][r][nh] = EL::mfma16(a[w % AD][mr], FRG ? fring[FRG ? (tap * 4 + q * 2 + nh) % FS : 0] : ring[FRG ? 0 : tap % 3][q * 2 + nh], acc[mr][r][nh]);
                    }
                    if (NP == 2) {
#pragma unroll
                        for (int nh = 0; nh < NHN; ++nh)
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr)
                                acc[mr][r][nh] = EL::mfma16(a[w % AD][mr], fring_lo[FRG ? (tap * 4 + q * 2 + nh) % FS : 0], acc[mr][r][nh]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
            HSIDM_SETPRIO(0);
            if (chunk == 0) HSIDM_STAMP(it, 4);
            lds_barrier();                                      // every wave is done reading: the tile may be overwritten
            if (chunk == 0) HSIDM_STAMP(it, 5);
        }
        if constexpr (PROJ) {
            // ---- the ResnetBlock's 1x1 projection of a second input (hsidm_conv_desc.ph[1]; reference unet.py:102-103,110) as up to THREE
            // more one-tap chunks accumulated into the same tile.  A projection chunk's raw 16 x 16 pixels are committed one halo row and
            // column further in than their place in the halo tile (commit_all), so the offsets of tap 8 - (2, 2) - read them; its weight
            // step comes through the same 3-step register ring.  The packed weights ALWAYS carry three projection steps per item (the host
            // pads with zero steps: ops.pack_layouts), so that the ring's phase is the same at every item start and every index stays
            // static; slots >= pchunks only pull their (zero) step through the ring.
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool live = j < p.pchunks;                // workgroup-uniform
                if (live) {
                    commit_all(true);
                    if (j == 0) HSIDM_STAMP(it, 6);             // (diagnostic builds: slots 6 .. 10 = the first projection chunk's phases)
                    stage_advance();
                    if (st_valid) {
                        if (st_chunk == 0) describe(st_item);
                        issue_all(st_chunk);
                    }
                    if (j == 0) HSIDM_STAMP(it, 7);
                    lds_barrier();
                    if (j == 0) HSIDM_STAMP(it, 8);
                }
                b_issue(ring[(j + 2) % 3]);
                if (live) {
                    HSIDM_SETPRIO(1);
                    x8 a[3][MR];
                    auto a_fetch = [&](int w) __attribute__((always_inline)) {      // w = 32 + 2 q + r: tap 8, slice q, row r of the pair
                        const int off = (2 + (w & 1)) * RP + 2 * PSTR + ((w >> 1) & 1) * 32;
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr) a[w % 3][mr] = *reinterpret_cast<const x8*>(halo + abase[mr] + off);
                    };
                    a_fetch(32);
                    a_fetch(33);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int w = 32 + kk, q = kk >> 1, r = kk & 1;
                        if (w + 2 < 36) a_fetch(w + 2);
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr)
                                acc[mr][r][nh] = EL::mfma16(a[w % 3][mr], ring[FRG ? 0 : j][q * 2 + nh], acc[mr][r][nh]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    HSIDM_SETPRIO(0);
                    if (j == 0) HSIDM_STAMP(it, 9);
                    lds_barrier();                              // every wave is done reading: the tile may be overwritten
                    if (j == 0) HSIDM_STAMP(it, 10);
                }
            }
        }
        HSIDM_STAMP(it, 12);

        // ---- epilogue (conv_v2's vector epilogue; the halo buffer is free between the two barriers) ------------------------
        if (HSIDM_ABL(64)) {                                    // (diagnostic builds: no epilogue; the accumulators stay live through one store)
            if (acc[0][0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][1][1] + acc[MR - 1][0][1][2];
            continue;
        }
        const int tr = tile - b * tiles_per_img;
        const int try_e = div_tx(tr);
        const int oy0 = try_e * TH, ox0 = (tr - try_e * p.tiles_x) * TW;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) ep_add[nh] += ep_bias[nh];
        if constexpr (SPL) {                                    // four consecutive couts per half: one 16-byte load each
            int lane_s = lane_id_now();
            asm volatile("" : "+v"(lane_s));
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                const int n4 = wn * 32 + 16 * nh + 4 * (lane_s >> 4);
                if (p.film) ep4[nh] = *reinterpret_cast<const f32x4*>(p.film + (size_t)b * p.film_stride + n4);
                if (p.bias) ep4[nh] += *reinterpret_cast<const f32x4*>(p.bias + n4);
            }
        }
        if constexpr (NCHW_) {
            // fp32 NCHW straight from the accumulator layout: lane = cout 16 nh + lane % 16 (3 live lanes), the four registers of
            // an accumulator = pixel columns 4 (lane / 16) .. + 3 of one tile row: one 16-byte store each
            int lane_o = lane_id_now();
            asm volatile("" : "+v"(lane_o));
            const int c_o = lane_o & 15, g_o = lane_o >> 4;
#pragma unroll
            for (int nh = 0; nh < NHN; ++nh) {
                const int n_o = 16 * nh + c_o;
                if (n_o < p.Cout) {
                    float* plane = reinterpret_cast<float*>(p.out) + ((size_t)b * p.Cout + n_o) * p.Hout * p.Wout;
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int oy = oy0 + wm * (16 / WM) + mr * 2 + r;
                            f32x4 o4;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o4[j] = fmaf(acc[mr][r][nh][j], kLn2, ep_add[nh]);
                            *reinterpret_cast<f32x4*>(plane + (size_t)oy * p.Wout + ox0 + 4 * g_o) = o4;
                        }
                }
            }
            HSIDM_STAMP(it, 13);
            continue;                                                               // no transposition patch: no barrier
        }
        E* scr = reinterpret_cast<E*>(smem_raw + PATCH_OFF) + wave * (32 * SCR_STR);
        int lane_e = lane_id_now();   // rebuilt here, not kept (conv_v2.h)
        asm volatile("" : "+v"(lane_e));
        const int pl0 = lane_e >> 2, cq = lane_e & 3;
        const int lc_e = lane_e & 15, lg_e = lane_e >> 4;
        const unsigned lane_el = (unsigned)(pl0 * p.Cout + cq * 8);                // pass pixel (row v4, column pl0): row part is uniform
        const size_t row_stride = (size_t)p.Wout * p.Cout;
        const size_t item_base = (((size_t)b * p.Hout + oy0 + wm * (16 / WM)) * p.Wout + ox0) * p.Cout + wn * 32;
        auto run = [&](auto leaky_tag, auto res_tag) __attribute__((always_inline)) {
            constexpr bool LEAKY = decltype(leaky_tag)::value != 0;
            constexpr bool RES = decltype(res_tag)::value != 0;
            x8 rv[RES ? 2 : 1];
            float vs1[8], vs2[8];
            // Without a residual the statistics are taken from the fp32 values before they are rounded for the store: the lane
            // owns one cout there, so 2 VALU per value and one exchange between the lane halves replace the unpacking of the
            // stored vectors and the 15-move butterfly (the rounding noise is zero-mean, 2^-9 relative: invisible to GroupNorm).
            // With a residual the stored sum only exists in the vector domain.
            float as1[2] = {0.f, 0.f}, as2[2] = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) vs1[k] = vs2[k] = 0.f;
#pragma unroll
            for (int g = 0; g < MR; ++g) {                                         // a pass = one MFMA tile = 2 tile rows x 16 columns
                // output address of the pass's vector v4: one 64-bit product chain per item (item_base), then multiples of the
                // row stride - the per-vector chain cost ~25 scalar instructions, 200 per epilogue
                auto vec_base = [&](int v4) __attribute__((always_inline)) -> size_t {
                    return item_base + (size_t)(g * 2 + v4) * row_stride;
                };
                if (RES) {
#pragma unroll
                    for (int v4 = 0; v4 < 2; ++v4) rv[v4] = *reinterpret_cast<const x8*>(reinterpret_cast<const E*>(p.res) + vec_base(v4) + lane_el);
                }
                if constexpr (SPL) {
                    // transposed accumulators: the lane holds couts 16 nh + 4 lg .. + 3 of pixel column lc -> ONE 8-byte patch write per
                    // (row, cout half) instead of four 2-byte ones
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) {
                            float v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                v[j] = fmaf(acc[g][r][nh][j], kLn2, ep4[nh][j]);
                                if (LEAKY) v[j] = v[j] > 0.f ? v[j] : 0.01f * v[j];
                            }
                            const x2 p0 = cvt_pair_hw<E>(v[0], v[1]), p1 = cvt_pair_hw<E>(v[2], v[3]);
                            u32x2 pk;
                            pk[0] = __builtin_bit_cast(unsigned, p0);
                            pk[1] = __builtin_bit_cast(unsigned, p1);
                            *reinterpret_cast<u32x2*>(scr + (16 * r + lc_e) * SCR_STR + 16 * nh + 4 * lg_e) = pk;
                        }
                } else {
                // patch row = 16 (row of the pair) + pixel column; the lane holds columns 4 lg .. + 3 of couts 16 nh + lc
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {                        // columns j, j + 1: one packed conversion (cvt_pair)
                            float v[2];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                v[e] = fmaf(acc[g][r][nh][j + e], kLn2, ep_add[nh]);   // the staged activations carry log2(e) (silu_log2e)
                                if (LEAKY) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
                                if (!RES) { as1[nh] += v[e]; as2[nh] = fmaf(v[e], v[e], as2[nh]); }   // statistics in the accumulator layout (see below)
                            }
                            const x2 pr = cvt_pair_hw<E>(v[0], v[1]);
                            const int at = (16 * r + 4 * lg_e + j) * SCR_STR + 16 * nh + lc_e;
                            scr[at] = pr[0];
                            scr[at + SCR_STR] = pr[1];
                        }
                }
#pragma unroll
                for (int v4 = 0; v4 < 2; ++v4) {
                    const x8 raw = *reinterpret_cast<const x8*>(scr + (pl0 + 16 * v4) * SCR_STR + cq * 8);
                    float f[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = (float)raw[k];
                    x8 o = raw;
                    if (RES) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            f[k] = fmaf(p.res_scale, f[k], (float)rv[v4][k]);   // statistics from the fp32 sum (the store's rounding noise is zero-mean)
                            o[k] = (E)f[k];
                        }
                    }
                    if (!HSIDM_ABL(1)) *reinterpret_cast<x8*>(reinterpret_cast<E*>(p.out) + vec_base(v4) + lane_el) = o;
                    if (RES || SPL) {                                           // (SPL: a lane of the accumulator layout holds eight couts - the vector domain is cheaper)
#pragma unroll
                        for (int k = 0; k < 8; ++k) { vs1[k] += f[k]; vs2[k] = fmaf(f[k], f[k], vs2[k]); }
                    }
                }
            }
            if (p.stats && !RES && !SPL) {
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {                                    // the four lanes lg = 0..3 of a cout hold its column quarters
                    float a = as1[nh] + lane_xor<16>(as1[nh], lane_e), d = as2[nh] + lane_xor<16>(as2[nh], lane_e);
                    a += lane_xor<32>(a, lane_e);
                    d += lane_xor<32>(d, lane_e);
                    if (lg_e == 0)
                        p.stats[((size_t)b * (tiles_per_img * WM) + tr * WM + wm) * p.Cout + wn * 32 + 16 * nh + lc_e] = make_float2(a, d);
                }
            }
            if (p.stats && (RES || SPL)) {                                          // one entry per (image, tile, pixel half)
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
                float* dst = reinterpret_cast<float*>(p.stats + ((size_t)b * (tiles_per_img * WM) + tr * WM + wm) * p.Cout + wn * 32);
                dst[(cq * 8 + (idx & 7)) * 2 + (idx >> 3)] = a1;
            }
        };
        if (p.act == ACT_LEAKY) { if (p.res) run(SlotTag<1>{}, SlotTag<1>{}); else run(SlotTag<1>{}, SlotTag<0>{}); }
        else                    { if (p.res) run(SlotTag<0>{}, SlotTag<1>{}); else run(SlotTag<0>{}, SlotTag<0>{}); }
        HSIDM_STAMP(it, 14);                                                        // no barrier: the patches do not alias the halo tile
        HSIDM_STAMP(it, 13);
    }
}

extern unsigned long long* g_stamps;      // conv_v2_inst.hip (diagnostic builds)

// Hout % 16 == 0, Wout % 16 == 0, transform = GN+SiLU, no upsampling (checked by the caller); nchw = 0: Cout == 64, NHWC bf16 out;
// nchw = 1: Cout <= 16 (the first 16-cout half of one padded 32-cout slice), fp32 NCHW out, no FiLM / residual / statistics
template <int WN_, bool NCHW_, typename E, int NP, bool SPL = false, bool PROJ = false>
static int launch_v3(ConvV2Params& p, int G, hipStream_t s) {
    static PerDeviceOnce once;
    if (int rc = raise_lds_cap(once, &conv_v3_kernel<WN_, NCHW_, E, NP, SPL, PROJ>, v3::LDS_BYTES)) return rc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_v3_kernel<WN_, NCHW_, E, NP, SPL, PROJ>), dim3(G), dim3(256), v3::LDS_BYTES, s, p);
    return (int)hipGetLastError();
}

// elem: 0 bf16, 1 fp16; np = 2 (fp16 only): p.w_lo holds the weights' low halves; spl (np = 2, NHWC form): p.w_ls / p.w_li hold them 2:4-compressed
int conv_v3_run(ConvV2Params& p, int nchw, int elem, int np, int spl, hipStream_t s) {
    const int g3_slots = 2 * device_cus();
    p.tiles_x = p.Wout / v3::TW;
    p.tiles_y = p.Hout / v3::TH;
    p.n_slices = 1;
    p.m_tiles = p.B * p.tiles_x * p.tiles_y;
    p.total_items = p.m_tiles;
    p.up_m = 0;
    const int no_xcd_map = debug_get(DBG_NO_XCD_MAP);
    p.xcd_m = (p.m_tiles % 8 == 0 && !no_xcd_map) ? 8 : 0;
    auto log2_or_neg = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return (1 << sh) == v ? sh : -1; };
    p.tpi_shift = log2_or_neg(p.tiles_x * p.tiles_y);
    p.tx_shift = log2_or_neg(p.tiles_x);
    p.abl = debug_get(DBG_V2_ABL);            // (read by diagnostic builds only: -DHSIDM_V2_ABLATE)
    p.stamps = g_stamps;
    int G = (p.total_items < g3_slots ? p.total_items : g3_slots) / 8 * 8;
    if (G == 0) G = p.total_items;
    if (p.pchunks > 0) {        // fused 1x1 projection (api.hip: v3_proj): the one-pass 64-cout forms
        if (nchw || np != 1 || p.pchunks > 3) return HSIDM_E_UNSUPPORTED;
        p.steps_per_item = 9 * p.nchunks + 3;                   // three projection steps per item in the packed weights, whatever pchunks (zero padded)
        return elem ? launch_v3<2, false, f16, 1, false, true>(p, G, s) : launch_v3<2, false, bf16, 1, false, true>(p, G, s);
    }
    if (elem == 0 && np == 1) return nchw ? launch_v3<1, true, bf16, 1>(p, G, s) : launch_v3<2, false, bf16, 1>(p, G, s);
    if (elem == 1 && np == 1) return nchw ? launch_v3<1, true, f16, 1>(p, G, s) : launch_v3<2, false, f16, 1>(p, G, s);
    if (elem == 1 && np == 2 && spl && !nchw) return launch_v3<2, false, f16, 2, true>(p, G, s);
    if (elem == 1 && np == 2) return nchw ? launch_v3<1, true, f16, 2>(p, G, s) : launch_v3<2, false, f16, 2>(p, G, s);
    return HSIDM_E_UNSUPPORTED;
}

}  // namespace hsidm
