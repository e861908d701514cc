// Instantiations + launcher of the persistent bf16 3x3 convolution (conv_v2.h).
#include "conv_v2.h"
#include <cstdlib>

namespace hsidm {

unsigned long long* g_stamps = nullptr;   // diagnostic builds only

int conv_v2_slots();

template <typename C>
static int run_v2(ConvV2Params& p, hipStream_t s) {
    constexpr size_t lds = C::LDS_BYTES;
    static_assert(C::NW == 8 ? lds <= 160 * 1024 : lds <= 80 * 1024, "two workgroups per CU (one with 8 waves)");
    static PerDeviceOnce once;
    if (int rc = raise_lds_cap(once, &conv_v2_kernel<C>, lds)) return rc;
    const int g_slots = conv_v2_slots();
    p.abl = debug_get(DBG_V2_ABL);
    p.stamps = g_stamps;
    const int imgs = (p.B + C::NI - 1) / C::NI;
    p.n_slices = p.Cout_pad / C::BN;
    const int tiles = imgs * p.tiles_x * p.tiles_y;
    p.m_tiles = C::UP4 ? 4 * tiles : tiles;
    p.up_m = 0;
    if (C::UP4 && 8 % p.n_slices == 0 && tiles % (8 / p.n_slices) == 0) p.up_m = 8 / p.n_slices;
    const int no_xcd_map = debug_get(DBG_NO_XCD_MAP);
    p.xcd_m = 0;
    if (!C::UP4 && !no_xcd_map && 8 % p.n_slices == 0 && p.m_tiles % (8 / p.n_slices) == 0) p.xcd_m = 8 / p.n_slices;
    auto log2_or_neg = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return (v > 0 && (1 << sh) == v) ? sh : -1; };
    p.ns_shift = log2_or_neg(p.n_slices);
    p.xm_shift = log2_or_neg(C::UP4 ? p.up_m : p.xcd_m);
    p.tpi_shift = log2_or_neg(p.tiles_x * p.tiles_y);
    p.tx_shift = log2_or_neg(p.tiles_x);
    p.total_items = p.m_tiles * p.n_slices;
    int lcm = 8;
    while (lcm % p.n_slices) lcm += 8;
    const int slots = C::NW == 8 ? g_slots / 2 : g_slots;
    int G = (p.total_items < slots ? p.total_items : slots) / lcm * lcm;
    if (G == 0) G = p.total_items;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_v2_kernel<C>), dim3(G), dim3(C::NTHR), lds, s, p);
    return (int)hipGetLastError();
}

void conv_v2_set_stamps(unsigned long long* p) { g_stamps = p; }

int conv_v2_subs(int tile_kind, int bn) {
    const int wm = 4 / (bn / 32);
    if (tile_kind == 2) return wm;                             // one-image 8x8 tile (bn = 128: one entry)
    return tile_kind == 0 ? wm : (wm >= 2 ? wm / 2 : 1);
}

int conv_v2_slots() { return 2 * device_cus(); }      // co-resident workgroups: 2 per CU

#define V2(BN, TH, TW, NI, XF) run_v2<V2Cfg<BN, TH, TW, NI, XF>>(p, s)

int conv_v2_run(int tile_kind, int bn, int xf, ConvV2Params& p, hipStream_t s) {
    if (xf == -2) {          // stride 2 over the four input-parity planes (weights: [plane][chunk][2x2 taps])
        if (bn == 128) return tile_kind == 0 ? run_v2<V2Cfg<128, 8, 16, 1, XF_NONE, 2>>(p, s) : run_v2<V2Cfg<128, 8, 8, 2, XF_NONE, 2>>(p, s);
        if (bn == 64) return tile_kind == 0 ? run_v2<V2Cfg<64, 8, 16, 1, XF_NONE, 2>>(p, s) : run_v2<V2Cfg<64, 8, 8, 2, XF_NONE, 2>>(p, s);
        return -2;
    }
    if (xf == -1) {          // parity-folded nearest-x2 (weights: 4 parities x [chunk][2x2 taps])
        if (bn != 128) return -2;
        return tile_kind == 0 ? run_v2<V2Cfg<128, 8, 16, 1, XF_NONE, 1>>(p, s) : run_v2<V2Cfg<128, 8, 8, 2, XF_NONE, 1>>(p, s);
    }
    if (xf != XF_NONE && xf != XF_AFFINE_SILU) return -2;
    const bool x = xf == XF_AFFINE_SILU;
    if (bn == 256) {         // 8 waves, one workgroup per CU: the staged tile is transformed once per 256 couts
        if (!x || tile_kind == 2) return -2;
        return tile_kind == 0 ? run_v2<V2Cfg<256, 8, 16, 1, XF_AFFINE_SILU, 0, 8>>(p, s) : run_v2<V2Cfg<256, 8, 8, 2, XF_AFFINE_SILU, 0, 8>>(p, s);
    }
    if (tile_kind == 2) return (x && bn == 128) ? V2(128, 8, 8, 1, XF_AFFINE_SILU) : -2;
    if (tile_kind == 0) {
        switch (bn) {
            case 128: return x ? V2(128, 8, 16, 1, XF_AFFINE_SILU) : V2(128, 8, 16, 1, XF_NONE);
            case 64: return x ? V2(64, 8, 16, 1, XF_AFFINE_SILU) : V2(64, 8, 16, 1, XF_NONE);
            case 32: return x ? V2(32, 8, 16, 1, XF_AFFINE_SILU) : V2(32, 8, 16, 1, XF_NONE);
        }
    } else {
        switch (bn) {
            case 128: return x ? V2(128, 8, 8, 2, XF_AFFINE_SILU) : V2(128, 8, 8, 2, XF_NONE);
            case 64: return x ? V2(64, 8, 8, 2, XF_AFFINE_SILU) : V2(64, 8, 8, 2, XF_NONE);
            case 32: return x ? V2(32, 8, 8, 2, XF_AFFINE_SILU) : V2(32, 8, 8, 2, XF_NONE);
        }
    }
    return -2;
}

}  // namespace hsidm
