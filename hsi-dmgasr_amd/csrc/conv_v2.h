// conv_v2: persistent implicit-GEMM 3x3 convolution (stride 1, optional nearest-x2 input) for the
// bf16 throughput mode.  Successor of conv_igemm.h for the shapes that carry ~90 % of the UNet's time.
//
// What changed against v1, and the measurement that motivated it (profiles/r01_baseline):
//   v1 spent 15 VALU + 5 SALU instructions per MFMA and its waves were parked 43 % of the time
//   (SQ_WAIT_ANY) behind one barrier + one LDS weight-tile commit per K step and an in-order vmcnt
//   that chained the short L2 weight loads to the long HBM halo loads.
//   * weights never touch LDS: each wave owns a 32-wide cout slice and streams its MFMA B-fragments
//     straight from L2 into registers (pre-packed so that one wave-load is one contiguous 1 KiB),
//     through a 3-deep register ring issued two K steps ahead -> no per-step barrier, no LDS writes;
//   * the input halo tile is double-buffered in LDS; the next chunk's tile is fetched at tap 0 and
//     normalised / SiLU'd / packed one 16-B vector per tap, in the shadow of the MFMAs; one LDS
//     barrier per chunk (9 K steps) instead of per step;
//   * the workgroup is persistent: it walks (pixel tile, cout slice) work items round-robin, so the
//     pipeline (weight ring, halo prefetch) runs across tile boundaries and tile-count quantisation
//     costs a fraction of a tile, not a launch wave; a block always works on the same cout slice, and
//     blocks that share an XCD (blockIdx % 8) share that slice -> each L2 streams 1/Nslices of W;
//   * the epilogue also emits per-(image, tile, channel) sum / sum-of-squares of what it stored, so
//     the consumer's GroupNorm needs no extra pass over the tensor (hsidm_gn_finalize reads the slab).
//
// Tile: 128 output pixels (8x16 of one image, or 8x8 of two) x BN couts, 4 waves; wave tile =
// (128/WM) pixels x 32 couts, WM x WN = 4, WN = BN/32.  K step = 64 input channels of one tap.
#pragma once
#include "conv_igemm.h"

namespace hsidm {

struct ConvV2Params {
    ConvPhase ph[2];
    int nphase;
    const bf16* w;          // packed [step][Cout_pad/32][kk 4][lane 64][8]
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
};

template <int BN_, int TH_, int TW_, int NI_, int XF_>
struct V2Cfg {
    static constexpr int BN = BN_, TH = TH_, TW = TW_, NI = NI_, XF = XF_;
    static constexpr int BM = 128, BK = 64;
    static_assert(TH * TW * NI == BM, "tile");
    static constexpr int WN = BN / 32, WM = 4 / WN, MR = BM / WM / 32;
    static constexpr int HROWS = TH + 2, HCOLS = TW + 2, HPIX = HROWS * HCOLS;
    static constexpr int PSTR = BK + 8, VPP = BK / 8;
    static constexpr int HVEC = NI * HPIX * VPP;
    static constexpr int MAXHV = (HVEC + 255) / 256;
    static constexpr int HALO_ELEMS = NI * HPIX * PSTR;
    static constexpr size_t LDS_BYTES = (size_t)2 * HALO_ELEMS * 2;
    // statistics sub-entries per spatial tile and image (see epilogue)
    static constexpr int SUBS = (NI == 1) ? WM : (WM >= 2 ? WM / 2 : 1);
};

template <int V> struct SlotTag { static constexpr int value = V; };

__device__ __forceinline__ void lds_barrier() {
    // LDS-only ordering: keeps the register-ring weight loads in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <typename C>
__global__ __launch_bounds__(256, 2) void conv_v2_kernel(const ConvV2Params p) {
    constexpr int BN = C::BN, TH = C::TH, TW = C::TW, NI = C::NI, MR = C::MR, WN = C::WN, WM = C::WM;
    constexpr int HPIX = C::HPIX, HCOLS = C::HCOLS, PSTR = C::PSTR, VPP = C::VPP, BK = C::BK;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* halo = reinterpret_cast<bf16*>(smem_raw);          // [2][HALO_ELEMS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int G = gridDim.x;
    const int tiles_per_img = p.tiles_x * p.tiles_y;

    int item = blockIdx.x;                                     // items: item = m_tile * n_slices + n_slice
    const int ns = item % p.n_slices;                          // constant for this block (G % n_slices == 0)
    const int n0 = ns * BN;
    const int n_items_blk = (p.total_items - item + G - 1) / G;
    const int total_steps = n_items_blk * p.steps_per_item;

    // ---- weight stream: one contiguous 1 KiB per (step, 32-cout slice, kk) --------------------------------
    const int nsw = p.Cout_pad >> 5;
    const bf16* wlane = p.w + ((size_t)(ns * WN + wn) * 4 * 64 + lane) * 8;
    const size_t wstep_stride = (size_t)nsw * 4 * 64 * 8;
    bf16x8 ring[3][4];
    int wnext = 0;                                              // step (within an item) of the next weight fetch
    auto b_issue = [&](bf16x8 (&dst)[4]) {
        const bf16* src = wlane + (size_t)wnext * wstep_stride;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dst[kk] = *reinterpret_cast<const bf16x8*>(src + kk * 64 * 8);
        wnext = (wnext + 1 == p.steps_per_item) ? 0 : wnext + 1;
    };

    // ---- halo staging state ---------------------------------------------------------------------------------
    const int cv = tid % VPP;
    int hv_lds[C::MAXHV];
#pragma unroll
    for (int i = 0; i < C::MAXHV; ++i) {
        const int v = tid + i * 256;
        hv_lds[i] = (v < C::HVEC) ? (v / VPP) * PSTR + cv * 8 : -1;
    }
    int hv_pix[C::MAXHV];                                       // for the tile being STAGED
    int st_b0 = 0;
    auto tile_coords = [&](int it, int& b0, int& oy0, int& ox0) __attribute__((always_inline)) {
        const int mt = it / p.n_slices;
        const int tg = mt / tiles_per_img;
        const int tr = mt - tg * tiles_per_img;
        b0 = tg * NI;
        oy0 = (tr / p.tiles_x) * TH;
        ox0 = (tr % p.tiles_x) * TW;
    };
    auto describe = [&](int it) __attribute__((always_inline)) {
        int b0, oy0, ox0;
        tile_coords(it, b0, oy0, ox0);
        st_b0 = b0;
#pragma unroll
        for (int i = 0; i < C::MAXHV; ++i) {
            const int v = tid + i * 256;
            int pix = -1;
            if (v < C::HVEC) {
                const int hp = v / VPP;
                const int img = hp / HPIX;
                const int r = hp - img * HPIX;
                const int hy = r / HCOLS, hx = r - hy * HCOLS;
                int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                const int b = b0 + img;
                bool ok = b < p.B && iy >= 0 && ix >= 0;
                if (p.ups) { ok = ok && iy < 2 * p.Hin && ix < 2 * p.Win; iy >>= 1; ix >>= 1; }
                else       { ok = ok && iy < p.Hin && ix < p.Win; }
                if (ok) pix = (b * p.Hin + iy) * p.Win + ix;
            }
            hv_pix[i] = pix;
        }
    };
    u32x4 hreg[C::MAXHV];
    float2 abreg[8];
    bool st_xf = false;
    int st_ctot = 0, st_c = 0;
    auto halo_issue = [&](const ConvPhase& ph, int chunk) __attribute__((always_inline)) {
        const int c = chunk * BK + cv * 8;
        const bf16* src;
        int cs, cl;
        if (c < ph.C0) { src = (const bf16*)ph.src0; cs = ph.C0; cl = c; }
        else           { src = (const bf16*)ph.src1; cs = ph.C1; cl = c - ph.C0; }
        st_ctot = ph.C0 + ph.C1;
        st_c = c;
        const bool cok = c < st_ctot;
#pragma unroll
        for (int i = 0; i < C::MAXHV; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            hreg[i] = (hv_pix[i] >= 0 && cok) ? *reinterpret_cast<const u32x4*>(src + (size_t)hv_pix[i] * cs + cl) : z;
        }
        st_xf = (C::XF != XF_NONE) && ph.transform != XF_NONE && cok;
        if (NI == 1 && st_xf) {
            const float2* ab = ph.gn_ab + (size_t)st_b0 * st_ctot + c;
#pragma unroll
            for (int k = 0; k < 8; ++k) abreg[k] = ab[k];
        }
    };
    const ConvPhase* st_ph = &p.ph[0];
    auto halo_commit_one = [&](int i, int buf) __attribute__((always_inline)) {
        if (hv_lds[i] < 0) return;
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(hreg[i][k] << 16);
            v[2 * k + 1] = __uint_as_float(hreg[i][k] & 0xffff0000u);
        }
        if (C::XF != XF_NONE) {
            if (st_xf && hv_pix[i] >= 0) {
                if (NI > 1) {
                    const int img = (hv_lds[i] / PSTR) / HPIX;
                    const float2* ab = st_ph->gn_ab + (size_t)(st_b0 + img) * st_ctot + st_c;
#pragma unroll
                    for (int k = 0; k < 8; ++k) abreg[k] = ab[k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = silu(fmaf(v[k], abreg[k].x, abreg[k].y));
            }
        }
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (bf16)v[k];
        *reinterpret_cast<bf16x8*>(halo + buf * C::HALO_ELEMS + hv_lds[i]) = o;
    };

    // ---- MFMA fragment bases --------------------------------------------------------------------------------------
    int abase[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int pm = wm * (C::BM / WM) + mr * 32 + lr;
        const int img = pm / (TH * TW);
        const int q = pm - img * (TH * TW);
        const int ty = q / TW, tx = q - ty * TW;
        abase[mr] = (img * HPIX + ty * HCOLS + tx) * PSTR + 8 * lh;
    }
    f32x16 acc[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mr][j] = 0.f;

    // ---- chunk bookkeeping -------------------------------------------------------------------------------------------
    const int nch0 = p.ph[0].nchunks;
    const int nch = nch0 + (p.nphase > 1 ? p.ph[1].nchunks : 0);
    int cur = 0;                 // halo buffer being consumed
    int chunk = 0, tap = 0;      // position inside the current item
    int ntaps = p.ph[0].ntaps;
    int st_item = item;          // item whose chunks are being staged
    int st_chunk = 0;
    bool st_valid = true;        // a chunk is pending in hreg / being staged

    // prologue: first weights, first halo tile (synchronously)
    b_issue(ring[0]);
    b_issue(ring[1]);
    describe(item);
    st_ph = &p.ph[0];
    halo_issue(*st_ph, 0);
#pragma unroll
    for (int i = 0; i < C::MAXHV; ++i) halo_commit_one(i, 0);
    lds_barrier();
    // advance the staging cursor to the chunk after (item, 0)
    auto stage_advance = [&]() __attribute__((always_inline)) {
        if (st_chunk + 1 < nch) { st_chunk += 1; }
        else { st_chunk = 0; st_item += G; }
        st_valid = st_item < p.total_items;
    };
    stage_advance();

    auto epilogue = [&](int it) __attribute__((always_inline)) {
        int b0, oy0, ox0;
        tile_coords(it, b0, oy0, ox0);
        const int mt = it / p.n_slices;
        const int trem = mt % tiles_per_img;
        const int n = n0 + wn * 32 + lr;
        const bool nok = n < p.Cout;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
        float s1[NI], s2[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) s1[q] = s2[q] = 0.f;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int pbase = wm * (C::BM / WM) + mr * 32;
            const int img = pbase / (TH * TW);                       // a 32-row MFMA tile never straddles images
            const int b = b0 + img;
            float film = 0.f;
            if (p.film && nok && b < p.B) film = p.film[(size_t)b * p.film_stride + n];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
                const int q = pbase + row - img * (TH * TW);
                const int ty = q / TW, tx = q - ty * TW;
                const int oy = oy0 + ty, ox = ox0 + tx;
                float v = acc[mr][j] + bias + film;
                acc[mr][j] = 0.f;
                if (!(nok && b < p.B && oy < p.Hout && ox < p.Wout)) continue;
                if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
                const size_t o = (((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + n;
                if (p.res) v = p.res_scale * v + (float)p.res[o];
                const bf16 st = (bf16)v;
                p.out[o] = st;
                const float sv = (float)st;
                s1[NI == 1 ? 0 : img] += sv;
                s2[NI == 1 ? 0 : img] += sv * sv;
            }
        }
        if (p.stats) {
            // wave partial over its pixels: combine the two lane halves, lanes 0..31 write one entry each
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                float a = s1[q] + __shfl_xor(s1[q], 32, 64);
                float d = s2[q] + __shfl_xor(s2[q], 32, 64);
                // which (image, sub-entry) this wave's sum belongs to
                int img, sub;
                if (NI == 1) { img = 0; sub = wm; }
                else if (WM == 1) { img = q; sub = 0; }
                else { img = (wm * (C::BM / WM)) / (TH * TW); sub = wm % (WM / 2 > 0 ? WM / 2 : 1); if (q != img) continue; }
                const int b = b0 + img;
                if (lh == 0 && nok && b < p.B)
                    p.stats[((size_t)b * (tiles_per_img * C::SUBS) + trem * C::SUBS + sub) * p.Cout + n] = make_float2(a, d);
            }
        }
    };

    // ---- one K step; SLOT = position in the weight ring (compile-time) ------------------------------------------------
    auto step = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_tag)::value;
        b_issue(ring[(SLOT + 2) % 3]);
        if (tap == 0 && st_valid) {                      // fetch the next chunk's halo vectors (and its GN params)
            if (st_chunk == 0) describe(st_item);
            st_ph = (st_chunk < nch0) ? &p.ph[0] : &p.ph[1];
            halo_issue(*st_ph, st_chunk < nch0 ? st_chunk : st_chunk - nch0);
        }
        int aoff = PSTR * (HCOLS + 1);                   // centre tap (fused 1x1 projection)
        if (ntaps == 9) {
            const int dy = tap / 3;
            aoff = (dy * HCOLS + (tap - 3 * dy)) * PSTR;
        }
        const bf16* hb = halo + cur * C::HALO_ELEMS + aoff;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[MR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) a[mr] = *reinterpret_cast<const bf16x8*>(hb + abase[mr] + kk * 16);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
                acc[mr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mr], ring[SLOT][kk], acc[mr], 0, 0, 0);
        }
        // staging work in the shadow of the MFMAs: vector (tap-2) now, everything left at the last tap
        if (st_valid) {
            const bool last_tap = tap == ntaps - 1;
#pragma unroll
            for (int i = 0; i < C::MAXHV; ++i) {
                const bool mine = (ntaps == 9) ? (tap == i + 2 || (last_tap && i + 2 > tap)) : true;
                if (mine) halo_commit_one(i, cur ^ 1);
            }
        }
        // advance
        tap += 1;
        if (tap == ntaps) {
            tap = 0;
            chunk += 1;
            lds_barrier();                               // next halo tile complete; this one free for re-use
            cur ^= 1;
            if (st_valid) stage_advance();
            if (chunk == nch) {
                epilogue(item);
                chunk = 0;
                item += G;
                ntaps = p.ph[0].ntaps;
            } else if (chunk == nch0) {
                ntaps = p.ph[1].ntaps;
            }
        }
    };

    for (int s = 0; s < total_steps; s += 3) {
        step(SlotTag<0>{});
        if (s + 1 >= total_steps) break;
        step(SlotTag<1>{});
        if (s + 2 >= total_steps) break;
        step(SlotTag<2>{});
    }
}

}  // namespace hsidm
