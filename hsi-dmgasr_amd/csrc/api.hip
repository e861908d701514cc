// C-ABI entry points: argument validation, tile selection and dispatch for the convolution.
#include "conv_v2.h"
#include <cstdlib>
#include <cstring>
#include "../../include/hsidm.h"

namespace hsidm {
#define HSIDM_DECL(tag) int tag(int tile_kind, int bn, const ConvParams& p, hipStream_t s);
HSIDM_DECL(conv_run_bf16_k3s1)
HSIDM_DECL(conv_run_bf16_k3s2)
HSIDM_DECL(conv_run_bf16_k1s1)
HSIDM_DECL(conv_run_bf16_k3s1nchw)
HSIDM_DECL(conv_run_f32x3_k3s1)
HSIDM_DECL(conv_run_f32x3_k3s2)
HSIDM_DECL(conv_run_f32x3_k1s1)
HSIDM_DECL(conv_run_f32x3_k3s1nchw)
HSIDM_DECL(conv_run_f16_k3s1)
HSIDM_DECL(conv_run_f16_k3s2)
HSIDM_DECL(conv_run_f16_k1s1)
HSIDM_DECL(conv_run_f16_k3s1nchw)
#undef HSIDM_DECL
int conv_v2_run_bf16(int tile_kind, int bn, int xf, ConvV2Params& p, hipStream_t s);
int conv_v2_run_f16(int tile_kind, int bn, int xf, ConvV2Params& p, hipStream_t s);
int conv_v2_run_f16w(int tile_kind, int bn, int xf, ConvV2Params& p, hipStream_t s);
int conv_v2_run_f32x3(int tile_kind, int bn, int xf, ConvV2Params& p, hipStream_t s);
int conv_v2_run_f32h(int tile_kind, int bn, int xf, ConvV2Params& p, hipStream_t s);
int conv_v2_subs(int tile_kind, int bn);
int conv_v2_slots();
int conv_v3_run(ConvV2Params& p, int nchw, int elem, int np, int spl, hipStream_t s);
void conv_v2_set_stamps(unsigned long long* p);
int conv_sk_parts(int B, int H, int W, int Cout, int nchunks);
int conv_sk_run(const bf16* src0, const bf16* src1, int C0, int C1, const float2* gn_ab, int silu, const bf16* w, const float* bias,
                const float* film, int film_stride, const bf16* res, float res_scale, bf16* out, float2* stats, int B, int H, int W,
                int Cout, int Cout_pad, int nchunks, const bf16* psrc0, const bf16* psrc1, int PC0, int PC1, int stride, float* workspace,
                int elem, hipStream_t s);
int conv1x1_g_run(int bn, int xf, const bf16* src0, const bf16* src1, int C0, int C1, const void* gn_ab, const bf16* w, const bf16* w_lo,
                  int elem, const float* bias, const bf16* res, float res_scale, bf16* out, float2* stats, int M, int HW, int Cout,
                  int nch, int im_H, int im_W, hipStream_t s);
}  // namespace hsidm

using namespace hsidm;

extern "C" int hsidm_version(void) { return HSIDM_ABI_VERSION; }

namespace hsidm {
int conv1x1_pair_run(const void* x, const bf16* w, const bf16* w_lo, int elem, const float* bias1, int act, const float* bias2, void* out,
                     float2* stats, int M, int HW, hipStream_t s);
}

extern "C" int hsidm_conv1x1_pair(int prec, const void* x, const void* w_pair, const void* w_pair_lo, const float* bias1, int act,
                                  const float* bias2, void* out, void* stats, int64_t M, int HW, void* stream) {
    if (!x || !w_pair || !w_pair_lo || !out || M <= 0 || HW <= 0 || M % HW || M > 0x7fffffff) return HSIDM_E_BADARG;
    if (prec != HSIDM_F16 && prec != HSIDM_F32X3) return HSIDM_E_UNSUPPORTED;
    if (act != HSIDM_ACT_NONE && act != HSIDM_ACT_LEAKY) return HSIDM_E_BADARG;
    if (HW % 64) return HSIDM_E_UNSUPPORTED;                    // 64-pixel statistics groups never straddle two images
    return hsidm::conv1x1_pair_run(x, reinterpret_cast<const hsidm::bf16*>(w_pair), reinterpret_cast<const hsidm::bf16*>(w_pair_lo),
                                   prec == HSIDM_F32X3 ? 2 : 1, bias1, act, bias2, out, reinterpret_cast<float2*>(stats), (int)M, HW,
                                   (hipStream_t)stream);
}

// Diagnostic builds (-DHSIDM_V2_STAMPS): device buffer [blocks][4][8][16] of s_memtime stamps; not part of hsidm.h.
extern "C" void hsidm_debug_set_stamps(void* p) { conv_v2_set_stamps(reinterpret_cast<unsigned long long*>(p)); }

extern "C" const char* hsidm_error_string(int code) {
    switch (code) {
        case HSIDM_OK: return "ok";
        case HSIDM_E_BADARG: return "hsidm: invalid argument";
        case HSIDM_E_UNSUPPORTED: return "hsidm: unsupported shape or mode";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "hsidm: unknown error";
    }
}

extern "C" int hsidm_conv_bk(int prec) {
    return (prec == HSIDM_BF16 || prec == HSIDM_F16 || prec == HSIDM_F32H) ? 64 : (prec == HSIDM_F32X3 ? 32 : HSIDM_E_BADARG);
}

// the 16-bit throughput modes share every kernel (templates over the element type, common.h: Elem)
static inline bool is16(int prec) { return prec == HSIDM_BF16 || prec == HSIDM_F16; }
// ... and the persistent 3x3 kernel also has an fp32 form (fp32 storage, bf16 hi + lo operands: conv_v2.h, AP = 2), taken when the
// descriptor carries both halves of the weights in the register-streaming order
// (HSIDM_F32H: fp32 storage, one fp16 activation operand, fp16 hi + lo weights - only the persistent forms exist)
static inline bool f32s(int prec) { return prec == HSIDM_F32X3 || prec == HSIDM_F32H; }
static inline bool v2_mode(const hsidm_conv_desc* d) { return is16(d->prec) || (f32s(d->prec) && d->w_v2 && d->w_v2_lo); }

enum { PATH_V1 = 0, PATH_V2 = 1, PATH_G1 = 3, PATH_V3 = 4, PATH_SK = 5 };

// Split-K form (conv_sk.hip): a bf16 3x3 stride-1 convolution whose persistent form would leave most workgroup slots empty
// (tiles x slices <= an eighth of them) while every item walks >= 4 channel chunks.  Returns the number of K parts, 0 = not eligible.
// A fused 1x1 projection (nphase == 2) is simply more chunks of one tap each: the split form is the only bf16 w_v2 kernel that
// takes it (the persistent kernels are single-phase), so the caller asks hsidm_conv_workspace_bytes first.
static int sk_parts(const hsidm_conv_desc* d, int Hout, int Wout) {
    const int xf = d->ph[0].transform;
    if (!is16(d->prec) || !d->w_v2 || d->w_v2_lo || d->out_nchw || d->ksize != 3 || d->ups ||
        (xf != HSIDM_XF_NONE && xf != HSIDM_XF_AFFINE_SILU) || d->bn != 128 || d->Cout % 128 || (Hout & 7) || (Wout & 7) ||
        d->act != HSIDM_ACT_NONE || debug_get(DBG_NO_SPLIT_K)) return 0;
    // stride 2 (Downsample): the parity-plane weights, even input maps, no transform, no projection
    if (d->stride == 2 && (xf != HSIDM_XF_NONE || d->nphase != 1 || (d->Hin & 1) || (d->Win & 1))) return 0;
    const int nchunks = (d->ph[0].C0 + d->ph[0].C1 + 63) / 64;
    if (nchunks < 4) return 0;
    const int TW = Wout >= 16 ? 16 : 8;
    const long long items = (long long)((d->B + (TW == 8)) / (TW == 8 ? 2 : 1)) * ((Wout + TW - 1) / TW) * ((Hout + 7) / 8) * (d->Cout / 128);
    // measured (profiles/r02_small_batch/sk_conv_bench.txt): the split form runs at ~400 TFLOP/s whatever the shape; the persistent
    // kernel passes that once it has ~80 work items (8x8 level at 40 latents: 55 us vs 61 us), and falls to 37-170 TFLOP/s below
    // 40 (8x8 / 16x16 levels at 5 latents: 81 -> 37 us, 72 -> 43 us)
    // (two-image 8x8 tiles keep the persistent kernel efficient down to fewer items than one-image 8x16 tiles do)
    // (stride 2 on 8x8 output maps: the persistent plane-wise kernel walks 16 window taps per chunk in sequence - 69 us at 40 latents
    // against 41 us split, profiles/r02_small_batch/sk_mult.txt; SK_MULT: threshold experiments)
    const int mult = debug_get(DBG_SK_MULT) > 0 ? debug_get(DBG_SK_MULT) : (TW == 8 ? (d->stride == 2 ? 4 : 8) : 4);
    if (items * mult > conv_v2_slots()) return 0;
    const int pchunks = d->nphase == 2 ? (d->ph[1].C0 + d->ph[1].C1 + 63) / 64 : 0;
    return conv_sk_parts(d->B, Hout, Wout, d->Cout, nchunks + pchunks);
}

// ---- diagnostic switches (common.h: DebugKey) ------------------------------------------------------
namespace {
struct DebugTable {
    std::atomic<int> v[hsidm::DBG_COUNT];
    DebugTable() {
        auto env_int = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
        v[hsidm::DBG_NO_V3] = getenv("HSIDM_NO_V3") ? 1 : 0;
        v[hsidm::DBG_V2_BN256] = env_int("HSIDM_V2_BN256", 1);
        v[hsidm::DBG_ATTENTION_V1] = env_int("HSIDM_ATTENTION_V1", 0);      // 1: the score-panel kernel, 2: attention_v2 (A/B against attention_v3)
        v[hsidm::DBG_NO_XCD_MAP] = getenv("HSIDM_NO_XCD_MAP") ? 1 : 0;
        const char* e = getenv("HSIDM_1X1");
        v[hsidm::DBG_1X1_V1] = (e && e[0] == 'v') ? 1 : 0;
        v[hsidm::DBG_V2_ABL] = env_int("HSIDM_V2_ABL", 0);
        v[hsidm::DBG_SK_MULT] = env_int("HSIDM_SK_MULT", 0);
        v[hsidm::DBG_NO_SPLIT_K] = getenv("HSIDM_NO_SPLIT_K") ? 1 : 0;
        v[hsidm::DBG_NO_SPARSE_LO] = getenv("HSIDM_NO_SPARSE_LO") ? 1 : 0;
        v[hsidm::DBG_NO_FUSED_PROJ] = getenv("HSIDM_NO_FUSED_PROJ") ? 1 : 0;
    }
};
DebugTable g_debug;          // constructed when the library is loaded
const char* const kDebugNames[hsidm::DBG_COUNT] = {"NO_V3", "V2_BN256", "ATTENTION_V1", "NO_XCD_MAP", "1X1_V1", "V2_ABL", "SK_MULT", "NO_SPLIT_K", "NO_SPARSE_LO", "NO_FUSED_PROJ"};
}  // namespace
int hsidm::debug_get(int key) { return g_debug.v[key].load(std::memory_order_relaxed); }

extern "C" int hsidm_debug_switch(const char* name, int value) {
    if (!name) return HSIDM_E_BADARG;
    for (int i = 0; i < hsidm::DBG_COUNT; ++i)
        if (!strcmp(name, kDebugNames[i])) {
            const int old = g_debug.v[i].exchange(value);
            return old < 0 ? 0 : old;
        }
    return HSIDM_E_BADARG;
}

extern "C" int hsidm_debug_query(const char* name) {
    if (!name) return HSIDM_E_BADARG;
    for (int i = 0; i < hsidm::DBG_COUNT; ++i)
        if (!strcmp(name, kDebugNames[i])) { const int v = g_debug.v[i].load(); return v < 0 ? 0 : v; }
    return HSIDM_E_BADARG;
}

static bool force_v1_1x1() { return debug_get(DBG_1X1_V1) == 1; }

static int conv_validate(const hsidm_conv_desc* d, int& Hout, int& Wout, int& tile_kind, int& path) {
    if (!d) return HSIDM_E_BADARG;
    if (d->prec != HSIDM_BF16 && d->prec != HSIDM_F32X3 && d->prec != HSIDM_F16 && d->prec != HSIDM_F32H) return HSIDM_E_BADARG;
    if (d->prec == HSIDM_F32H && (!d->w_v2 || !d->w_v2_lo)) return HSIDM_E_BADARG;
    if (d->w_v2_lo && (d->prec == HSIDM_BF16 || !d->w_v2)) return HSIDM_E_BADARG;
    if ((d->w_v2_ls || d->w_v2_li) && (!d->w_v2_lo || !d->w_v2_ls || !d->w_v2_li || d->prec != HSIDM_F16)) return HSIDM_E_BADARG;
    if (d->nphase < 1 || d->nphase > 2) return HSIDM_E_BADARG;
    if (d->ksize != 3 && d->ksize != 1) return HSIDM_E_BADARG;
    if (d->stride != 1 && d->stride != 2) return HSIDM_E_BADARG;
    if (d->bn != 32 && d->bn != 64 && d->bn != 128) return HSIDM_E_BADARG;
    if (d->B <= 0 || d->Hin <= 0 || d->Win <= 0 || d->Cout <= 0) return HSIDM_E_BADARG;
    if (d->ups < 0 || d->ups > HSIDM_UPS_FOLDED) return HSIDM_E_BADARG;
    // geometry the reference uses: 3x3 pad 1 (stride 1|2, optional nearest x2), 1x1 pad 0 stride 1
    if (d->ksize == 1 && (d->stride != 1 || d->ups)) return HSIDM_E_UNSUPPORTED;
    if (d->ups && d->stride != 1) return HSIDM_E_UNSUPPORTED;
    if (d->out_nchw && (d->ksize != 3 || d->stride != 1 || d->bn == 64 || d->stats)) return HSIDM_E_UNSUPPORTED;
    if (d->nphase == 2 && (d->ksize != 3 || d->stride != 1 || d->ups || d->ph[1].ntaps != 1)) return HSIDM_E_UNSUPPORTED;
    // a fused projection reads its input RAW on every kernel that takes one (ResnetBlock.res_conv(x), reference unet.py:110): a
    // transformed projection phase would give path-dependent results (the LDS-tiled kernel would apply it, conv_v3 / conv_sk not)
    if (d->nphase == 2 && d->ph[1].transform != HSIDM_XF_NONE) return HSIDM_E_UNSUPPORTED;
    Hout = d->ups ? 2 * d->Hin : (d->stride == 2 ? (d->Hin + 1) / 2 : d->Hin);
    Wout = d->ups ? 2 * d->Win : (d->stride == 2 ? (d->Win + 1) / 2 : d->Win);
    if (Hout != d->Hout || Wout != d->Wout) return HSIDM_E_BADARG;
    // spatial tile: 8x16 pixels of one image, or 8x8 pixels of two images when the map is narrow
    // (parity-folded upsample: tiles live on the INPUT grid)
    const bool up4 = d->ups == HSIDM_UPS_FOLDED;
    tile_kind = ((up4 ? d->Win : Wout) >= 16) ? 0 : 1;
    const int xf = d->ph[0].transform;
    path = PATH_V1;
    if (up4) {
        if (!v2_mode(d) || !d->w_v2 || d->out_nchw || d->nphase != 1 || d->ksize != 3 || xf != HSIDM_XF_NONE ||
            d->bn != 128) return HSIDM_E_UNSUPPORTED;
    }
    // stride 2 with w_v2: the four input-parity planes on the conv_v2 schedule (w_v2 = [plane][chunk][2x2 taps] layout)
    if (v2_mode(d) && d->w_v2 && !d->out_nchw && d->nphase == 1 && d->stride == 2) {
        if (d->ksize != 3 || xf != HSIDM_XF_NONE || (d->bn != 64 && d->bn != 128) || (d->Hin & 1) || (d->Win & 1)) return HSIDM_E_UNSUPPORTED;
        path = PATH_V2;
    }
    // fp32 NCHW output (the UNet's final Block, 64 -> 3; Cout <= 16): the first 16-cout half of one padded 32-cout slice on the 256-pixel kernel (conv_v3.hip, WN = 1)
    if (is16(d->prec) && d->w_v2 && d->out_nchw && d->nphase == 1 && d->stride == 1 && d->ksize == 3 && !d->ups &&
        xf == HSIDM_XF_AFFINE_SILU && d->bn == 32 && d->Cout <= 16 && Hout % 16 == 0 && Wout % 16 == 0 && !d->film && !d->res &&
        !d->stats && d->act == HSIDM_ACT_NONE && !debug_get(DBG_NO_V3)) path = PATH_V3;
    // A fused 1x1 projection (nphase == 2) on a persistent kernel: conv_v3's one-pass 64-cout forms (whole 16x16 tiles) walk it as up to
    // three more one-tap chunks; everything else with a projection stays on the split-K or the LDS-tiled kernel
    // (one-pass weights only - bf16, the fp16 policy's dithered sets - and at most three projection chunks; a layer with hi + lo weights keeps
    // its projection a launch of its own: round 4's sparse-lo projection form was retired with the policy that needed it)
    const bool v3_proj = d->nphase == 2 && is16(d->prec) && d->w_v2 && !d->w_v2_lo && d->ph[1].C0 + d->ph[1].C1 <= 192 && !d->out_nchw &&
                         d->stride == 1 && d->ksize == 3 && !d->ups && xf == HSIDM_XF_AFFINE_SILU && d->bn == 64 && d->Cout == 64 &&
                         Hout % 16 == 0 && Wout % 16 == 0 && !d->res && (d->ph[0].C0 + d->ph[0].C1) % 64 == 0 &&
                         !debug_get(DBG_NO_V3) && !debug_get(DBG_NO_FUSED_PROJ);
    if (v3_proj) path = PATH_V3;
    if (v2_mode(d) && d->w_v2 && !d->out_nchw && d->nphase == 1 && d->stride == 1) {
        if (d->ksize == 3 && (xf == HSIDM_XF_NONE || xf == HSIDM_XF_AFFINE_SILU)) path = PATH_V2;
        // (fp32 mode: a shape its two persistent forms - this kernel and the 1x1 GEMM - do not take stays on the LDS-tiled kernel)
        // 8x8 maps: two-image tiles halve the work items; when those would leave half of the co-resident workgroup slots
        // empty, one-image 64-pixel tiles keep two workgroups on every CU at the same staging cost per pixel
        if (path == PATH_V2 && tile_kind == 1 && xf == HSIDM_XF_AFFINE_SILU && !d->ups && d->bn == 128 && Hout <= 8 && Wout <= 8 &&
            (long long)d->B * ((d->Cout + 127) / 128) <= conv_v2_slots()) tile_kind = 2;
        // 64-cout GN+SiLU layers on whole 16x16 tiles: the 256-pixel kernel (conv_v3.hip); HSIDM_NO_V3=1: diagnostic A/B switch
        if (path == PATH_V2 && is16(d->prec) && xf == HSIDM_XF_AFFINE_SILU && !d->ups && d->bn == 64 && d->Cout == 64 && Hout % 16 == 0 &&
            Wout % 16 == 0 && !debug_get(DBG_NO_V3)) path = PATH_V3;
        // 8 input channels (one 16-byte vector per pixel): w_v2 is the tap-major GEMM layout (include/hsidm.h), which only
        // the GEMM kernel reads
        if (is16(d->prec) && d->ksize == 3 && d->ph[0].C0 + d->ph[0].C1 == 8) {
            const bool g1 = !d->ups && xf == HSIDM_XF_NONE && d->act == HSIDM_ACT_NONE && !d->film && d->ph[0].C1 == 0 &&
                            (d->bn == 64 || d->bn == 128) && d->Cout % d->bn == 0 && (Hout * Wout) % 64 == 0 &&
                            (Wout & (Wout - 1)) == 0 && Hout * Wout >= 128;
            if (!g1) return HSIDM_E_UNSUPPORTED;
            path = PATH_G1;
        }
        // LDS-staged GEMM (conv1x1_g.hip): whole cout slices, 64-pixel statistics groups
        if (d->ksize == 1 && (xf == HSIDM_XF_NONE || xf == HSIDM_XF_AFFINE) && d->act == HSIDM_ACT_NONE && !d->film &&
            (d->bn == 64 || d->bn == 128) && d->Cout % d->bn == 0 && (Hout * Wout) % 64 == 0 && !force_v1_1x1()) path = PATH_G1;
    }
    // HSIDM_F32H has the 8x16-tile forms of the persistent 3x3 kernel and the plain GEMM, nothing else (two-image tiles and the GEMM's
    // GroupNorm prologue spill at two workgroups per CU with 8-byte staging registers): the caller runs every other shape with
    // HSIDM_F32X3 weights - same tensors in, same tensors out
    if (d->prec == HSIDM_F32H && !((path == PATH_V2 && tile_kind == 0) || (path == PATH_G1 && xf == HSIDM_XF_NONE))) return HSIDM_E_UNSUPPORTED;
    if ((path == PATH_V2 || d->nphase == 2) && d->workspace) {
        const int parts = sk_parts(d, Hout, Wout);
        if (parts > 0 && d->workspace_bytes >= (int64_t)parts * d->B * Hout * Wout * d->Cout * 4) path = PATH_SK;
    }
    return HSIDM_OK;
}

// Cout slice a conv_v2 launch works on.  Packed for 128-cout slices, a GroupNorm+SiLU conv whose Cout is a multiple of 256 runs
// 256-cout items on 8 waves (one workgroup per CU, conv_v2.h NW = 8): the staged halo tile is transformed once per 256 couts
// instead of once per 128 (measured at batch 240: 16x16 level +7 %, 8x8 level +11 %, 32x32 level +1..2 %, step +1.5 %).
// Only when that still occupies half of the CUs: the items are half as many and the slots are one per CU.  HSIDM_V2_BN256=0 disables it.
static int v2_slice(const hsidm_conv_desc* d, int Hout, int Wout, int tile_kind, int path) {
    const int on = debug_get(DBG_V2_BN256);
    if (!on || path != PATH_V2 || tile_kind == 2 || d->bn != 128 || d->stride != 1 || d->ups || d->w_v2_lo ||
        d->ph[0].transform != HSIDM_XF_AFFINE_SILU || d->Cout % 256) return d->bn;
    const int TW = tile_kind == 0 ? 16 : 8;
    const long long tiles = (long long)((d->B + (tile_kind == 1)) / (tile_kind == 1 ? 2 : 1)) * ((Wout + TW - 1) / TW) * ((Hout + 7) / 8);
    const long long items256 = tiles * (d->Cout / 256);
    const int cus = conv_v2_slots() / 2;
    if (items256 < cus / 2) return d->bn;                                      // at least half of the CUs get an item
    if (on == 2) return 256;                                                   // (A/B: the rule without the makespan comparison)
    if (items256 <= cus) return 256;                                           // one round either way: the form with less staging
    // More than one round of 256-cout items: the last, partly filled round costs a whole item.  128-cout items fill the tail
    // better - two are resident per CU, each 1.07 item-times when it shares the CU and ~0.6 when it has the CU to itself
    // (measured: +7 % for the 128-cout form at full occupancy, conv_bench at batch 40) - so compare the two makespans, in units
    // of one 256-cout item (40 latents, 32x32 level: 320 items = 2 rounds against 1.07 + 0.6; 240 latents: 8 against 8.1).
    const double t256 = (double)((items256 + cus - 1) / cus);
    const long long items128 = 2 * items256, full = items128 / (2 * cus), rem = items128 % (2 * cus);
    const double t128 = 1.07 * (double)full + (rem == 0 ? 0.0 : (rem <= cus ? 0.6 : 1.07));
    return t256 <= t128 ? 256 : d->bn;
}

extern "C" int64_t hsidm_conv_workspace_bytes(const hsidm_conv_desc* d) {
    int Hout, Wout, tile_kind, path;
    hsidm_conv_desc probe = *d;
    probe.workspace = nullptr;
    const int rc = conv_validate(&probe, Hout, Wout, tile_kind, path);
    if (rc != HSIDM_OK) return rc;
    if (path != PATH_V2 && d->nphase != 2) return 0;
    const int parts = sk_parts(d, Hout, Wout);
    return (int64_t)parts * d->B * Hout * Wout * d->Cout * 4;
}

extern "C" int hsidm_conv_kernel_id(const hsidm_conv_desc* d) {
    int Hout, Wout, tile_kind, path;
    const int rc = conv_validate(d, Hout, Wout, tile_kind, path);
    if (rc != HSIDM_OK) return rc;
    /* tile kinds: 0 8x16, 1 8x8 of two images, 2 8x8 of one; bits 8.. = couts per work item */
    if (path == PATH_SK) return path | (2 << 4) | (128 << 8);          /* 8x8 tiles of one image, 128 couts per workgroup */
    return path | ((path == PATH_G1 ? 0 : tile_kind) << 4) | (v2_slice(d, Hout, Wout, tile_kind, path) << 8);
}

extern "C" int hsidm_conv_stats_nsplit(const hsidm_conv_desc* d) {
    int Hout, Wout, tile_kind, path;
    const int rc = conv_validate(d, Hout, Wout, tile_kind, path);
    if (rc != HSIDM_OK) return rc;
    if (d->out_nchw) return HSIDM_E_UNSUPPORTED;
    if (path == PATH_V3) return (Hout / 16) * (Wout / 16) * 2;
    if (path == PATH_G1 || path == PATH_SK) return Hout * Wout / 64;
    const bool use_v2 = path == PATH_V2;
    const int TW = tile_kind == 0 ? 16 : 8;
    if (d->ups == HSIDM_UPS_FOLDED)      // one entry per (input tile, parity, wave row)
        return ((d->Win + TW - 1) / TW) * ((d->Hin + 7) / 8) * 4 * conv_v2_subs(tile_kind, d->bn);
    const int tiles = ((Wout + TW - 1) / TW) * ((Hout + 7) / 8);
    // v1 and v2 use the same wave grid rule per cout slice: WN = 2 (v1, bn >= 64) / bn/32 (v2)
    int subs;
    if (use_v2) subs = conv_v2_subs(tile_kind, d->bn);
    else {
        const int wm = d->bn >= 64 ? 2 : 4;
        subs = tile_kind == 0 ? wm : (wm >= 2 ? wm / 2 : 1);
    }
    return tiles * subs;
}

extern "C" int hsidm_conv2d(const hsidm_conv_desc* d, void* stream) {
    int Hout, Wout, tile_kind, path;
    const int rc = conv_validate(d, Hout, Wout, tile_kind, path);
    if (rc != HSIDM_OK) return rc;
    const bool use_v2 = path == PATH_V2 || path == PATH_V3;
    if (!d->out || (!d->w_hi && d->prec != HSIDM_F32H)) return HSIDM_E_BADARG;       // (HSIDM_F32H: only w_v2 / w_v2_lo are read)
    if (d->prec == HSIDM_F32X3 && !use_v2 && path != PATH_G1 && !d->w_lo) return HSIDM_E_BADARG;      // (the persistent forms read w_v2 / w_v2_lo)
    const int elem = d->prec == HSIDM_F16 ? 1 : 0;
    const int bk = (use_v2 || path == PATH_G1) ? 64 : hsidm_conv_bk(d->prec);        // (the persistent kernels walk 64-channel chunks in every mode)
    ConvParams p;
    p.nphase = d->nphase;
    int steps = 0;
    for (int i = 0; i < 2; ++i) {
        ConvPhase& q = p.ph[i];
        if (i < d->nphase) {
            const hsidm_conv_phase& s = d->ph[i];
            if (!s.src0 || s.C0 <= 0 || (s.C0 & 7) || s.C1 < 0 || (s.C1 & 7) || (s.C1 > 0 && !s.src1)) return HSIDM_E_BADARG;
            if (s.transform != HSIDM_XF_NONE && !s.gn_ab) return HSIDM_E_BADARG;
            if (s.ntaps != (i == 0 ? d->ksize * d->ksize : 1)) return HSIDM_E_BADARG;
            q.src0 = s.src0; q.src1 = s.C1 > 0 ? s.src1 : nullptr;
            q.gn_ab = reinterpret_cast<const float2*>(s.gn_ab);
            q.C0 = s.C0; q.C1 = s.C1; q.transform = s.transform; q.ntaps = s.ntaps;
            q.nchunks = (s.C0 + s.C1 + bk - 1) / bk;
            steps += q.nchunks * q.ntaps;
        } else {
            q.src0 = q.src1 = nullptr; q.gn_ab = nullptr; q.C0 = q.C1 = 0; q.transform = 0; q.ntaps = 0; q.nchunks = 0;
        }
    }
    const int TH = 8, TW = tile_kind == 0 ? 16 : 8;
    const bool up4 = d->ups == HSIDM_UPS_FOLDED;
    const int tiles_x = ((up4 ? d->Win : Wout) + TW - 1) / TW, tiles_y = ((up4 ? d->Hin : Hout) + TH - 1) / TH;
    const int cout_pad = (d->Cout + d->bn - 1) / d->bn * d->bn;
    hipStream_t s = (hipStream_t)stream;
    if (path == PATH_G1) {
        const hsidm_conv_phase& s0 = d->ph[0];
        return conv1x1_g_run(d->bn, s0.transform, reinterpret_cast<const bf16*>(s0.src0), reinterpret_cast<const bf16*>(s0.C1 > 0 ? s0.src1 : nullptr),
                             s0.C0, s0.C1, s0.gn_ab, reinterpret_cast<const bf16*>(d->w_v2), reinterpret_cast<const bf16*>(d->w_v2_lo),
                             d->prec == HSIDM_F32X3 ? 2 : (d->prec == HSIDM_F32H ? 3 : elem),
                             d->bias, reinterpret_cast<const bf16*>(d->res),
                             d->res_scale, reinterpret_cast<bf16*>(d->out), reinterpret_cast<float2*>(d->stats), d->B * Hout * Wout,
                             Hout * Wout, d->Cout, d->ksize == 3 ? 2 : (s0.C0 + s0.C1 + 127) / 128 * 2, d->ksize == 3 ? Hout : 0,
                             d->ksize == 3 ? Wout : 0, s);
    }
    if (path == PATH_SK) {
        const hsidm_conv_phase& s0 = d->ph[0];
        return conv_sk_run(reinterpret_cast<const bf16*>(s0.src0), reinterpret_cast<const bf16*>(s0.C1 > 0 ? s0.src1 : nullptr), s0.C0, s0.C1,
                           reinterpret_cast<const float2*>(s0.transform != HSIDM_XF_NONE ? s0.gn_ab : nullptr),
                           s0.transform == HSIDM_XF_AFFINE_SILU, reinterpret_cast<const bf16*>(d->w_v2), d->bias, d->film, d->film_stride,
                           reinterpret_cast<const bf16*>(d->res), d->res_scale, reinterpret_cast<bf16*>(d->out),
                           reinterpret_cast<float2*>(d->stats), d->B, Hout, Wout, d->Cout, cout_pad, p.ph[0].nchunks,
                           reinterpret_cast<const bf16*>(p.ph[1].src0), reinterpret_cast<const bf16*>(p.ph[1].src1), p.ph[1].C0, p.ph[1].C1,
                           d->stride, reinterpret_cast<float*>(d->workspace), elem, s);
    }
    if (use_v2) {
        ConvV2Params v{};
        v.src0 = reinterpret_cast<const bf16*>(p.ph[0].src0);
        v.src1 = reinterpret_cast<const bf16*>(p.ph[0].src1);
        v.gn_ab = reinterpret_cast<const f32x4*>(p.ph[0].gn_ab);
        v.C0 = p.ph[0].C0; v.C1 = p.ph[0].C1; v.nchunks = p.ph[0].nchunks;
        v.w = reinterpret_cast<const bf16*>(d->w_v2);
        v.w_lo = reinterpret_cast<const bf16*>(d->w_v2_lo);
        const bool sparse_ok = d->w_v2_ls && d->w_v2_li && !debug_get(DBG_NO_SPARSE_LO);      // HSIDM_NO_SPARSE_LO=1: diagnostic A/B switch
        v.w_ls = sparse_ok ? reinterpret_cast<const bf16*>(d->w_v2_ls) : nullptr;
        v.w_li = sparse_ok ? reinterpret_cast<const int*>(d->w_v2_li) : nullptr;
        const int np = d->w_v2_lo ? 2 : 1;
        auto v2_run = [&](int tk, int bn_, int xf_) {
            if (d->prec == HSIDM_F32X3) return conv_v2_run_f32x3(tk, bn_, xf_, v, s);
            if (d->prec == HSIDM_F32H) return conv_v2_run_f32h(tk, bn_, xf_, v, s);
            if (!elem) return conv_v2_run_bf16(tk, bn_, xf_, v, s);
            return np == 2 ? conv_v2_run_f16w(tk, bn_, xf_, v, s) : conv_v2_run_f16(tk, bn_, xf_, v, s);
        };
        v.bias = d->bias; v.film = d->film; v.film_stride = d->film_stride;
        v.res = reinterpret_cast<const bf16*>(d->res); v.res_scale = d->res_scale;
        v.out = reinterpret_cast<bf16*>(d->out); v.stats = reinterpret_cast<float2*>(d->stats);
        v.B = d->B; v.Hin = d->Hin; v.Win = d->Win; v.Hout = Hout; v.Wout = Wout; v.Cout = d->Cout; v.Cout_pad = cout_pad;
        v.ups = d->ups; v.act = d->act; v.tiles_x = tiles_x; v.tiles_y = tiles_y;
        if (d->nphase == 2) {                    // (conv_validate: only the forms that take a projection get here with one)
            v.psrc0 = reinterpret_cast<const bf16*>(p.ph[1].src0);
            v.psrc1 = reinterpret_cast<const bf16*>(p.ph[1].src1);
            v.PC0 = p.ph[1].C0; v.PC1 = p.ph[1].C1; v.pchunks = p.ph[1].nchunks;
        }
        if (path == PATH_V3) {
            v.steps_per_item = steps;
            // the second weight pass as a 2:4 structured-sparse one when the caller packed it (HSIDM_NO_SPARSE_LO=1: diagnostic A/B switch)
            const int spl = np == 2 && sparse_ok && !d->out_nchw;
            return conv_v3_run(v, d->out_nchw, elem, np, spl, s);
        }
        const bool dn4 = d->stride == 2;
        if (dn4) v.nchunks = 4 * p.ph[0].nchunks;
        v.steps_per_item = dn4 ? v.nchunks * 4 : (up4 ? p.ph[0].nchunks * 4 : steps);
        if (!dn4 && !up4 && v2_slice(d, Hout, Wout, tile_kind, path) == 256) return v2_run(tile_kind, 256, d->ph[0].transform);
        return v2_run(tile_kind, d->bn, dn4 ? -2 : (up4 ? -1 : d->ph[0].transform));
    }
    p.w_hi = reinterpret_cast<const bf16*>(d->w_hi);
    p.w_lo = reinterpret_cast<const bf16*>(d->w_lo);
    p.bias = d->bias; p.film = d->film; p.film_stride = d->film_stride;
    p.res = d->res; p.res_scale = d->res_scale; p.out = d->out;
    p.B = d->B; p.Hin = d->Hin; p.Win = d->Win; p.Hout = Hout; p.Wout = Wout; p.Cout = d->Cout;
    p.Cout_pad = cout_pad;
    p.ups = d->ups; p.act = d->act;
    p.stats = reinterpret_cast<float2*>(d->stats);
    p.tiles_x = tiles_x;
    p.tiles_y = tiles_y;
    if (d->prec == HSIDM_F16) {                  // the generic kernel's fp16 form always multiplies by hi + lo weights
        if (!d->w_lo) return HSIDM_E_BADARG;
        if (d->out_nchw) return conv_run_f16_k3s1nchw(tile_kind, d->bn, p, s);
        if (d->ksize == 1) return conv_run_f16_k1s1(tile_kind, d->bn, p, s);
        if (d->stride == 2) return conv_run_f16_k3s2(tile_kind, d->bn, p, s);
        return conv_run_f16_k3s1(tile_kind, d->bn, p, s);
    }
    const bool bf = d->prec == HSIDM_BF16;
    if (d->out_nchw) return bf ? conv_run_bf16_k3s1nchw(tile_kind, d->bn, p, s) : conv_run_f32x3_k3s1nchw(tile_kind, d->bn, p, s);
    if (d->ksize == 1) return bf ? conv_run_bf16_k1s1(tile_kind, d->bn, p, s) : conv_run_f32x3_k1s1(tile_kind, d->bn, p, s);
    if (d->stride == 2) return bf ? conv_run_bf16_k3s2(tile_kind, d->bn, p, s) : conv_run_f32x3_k3s2(tile_kind, d->bn, p, s);
    return bf ? conv_run_bf16_k3s1(tile_kind, d->bn, p, s) : conv_run_f32x3_k3s1(tile_kind, d->bn, p, s);
}
