// conv1x1_ws: weight-stationary 1x1 convolution (a TN GEMM: out[pixel][cout] = x[pixel][:] . W[cout][:]) for the
// bf16 throughput mode.  Used for the ResnetBlock residual projections (reference unet.py:102-103) and the attention
// output projection (unet.py:141) - 24 launches per UNet call that the generic v1 kernel ran at ~220 TFLOP/s.
//
// Why a different shape from conv_v2: a 1x1 convolution has no tap reuse, so staging the activation tile through LDS
// costs a barrier per 16 MFMAs.  Here the roles are swapped:
//   * WEIGHTS are the stationary operand: the whole [K][BN] slice of this workgroup (<= 64 KiB, already in MFMA
//     B-fragment order) is copied to LDS once; every wave then reads conflict-free 1 KiB fragments from it;
//   * ACTIVATIONS stream: a wave owns 32 pixels at a time and loads their rows straight from global memory as
//     MFMA A-fragments (16 B per lane), eight k-slices ahead through a register ring that runs across tile
//     boundaries; no barrier anywhere in the main loop;
//   * 16 waves (1024 threads) per workgroup share one weight copy, one workgroup per CU; workgroups that share an XCD
//     (blockIdx % 8) serve the same cout slices;
//   * epilogue as in conv_v2: per-wave LDS transposition, 16-byte stores, optional residual and statistics slab.
#include "conv_v2.h"
#include "../../include/hsidm.h"

namespace hsidm {

template <int BN>
__global__ __launch_bounds__(1024) void conv1x1_ws_kernel(const C1Params p) {
    constexpr int NR = BN / 32;
    constexpr int D = 8;                                    // A-fragment lookahead (k-slices of 16)
    constexpr int SCR_STR = 40;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* wl = reinterpret_cast<bf16*>(smem_raw);           // [ksteps][NR][4][64][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    // Workgroups b and b + 8 sit on the same XCD (round-robin dispatch; speed only, never correctness).  The cout slices
    // of one pixel range are given to such workgroups, so the second slice re-reads its activation rows from that
    // XCD's L2 instead of HBM (the weights are in LDS and do not compete for it).
    int slice, brank;
    if (p.blocks_per_slice % 8 == 0) {
        slice = (blockIdx.x >> 3) % p.n_slices;
        brank = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * p.n_slices));
    } else {
        slice = blockIdx.x % p.n_slices;
        brank = blockIdx.x / p.n_slices;
    }
    const int n0 = slice * BN;
    const int U = p.ksteps * 4;                             // k-slices per tile

    // ---- weights -> LDS (fragment order, conflict-free 16-B lanes) ----------------------------------------------
    {
        const int nsw = p.Cout_pad >> 5;
        const int per_step = NR * 4 * 64;                   // 16-B vectors per k step in this slice
        for (int v = tid; v < p.ksteps * per_step; v += 1024) {
            const int ks = v / per_step, r = v - ks * per_step;
            const u32x4 val = *reinterpret_cast<const u32x4*>(p.w + (((size_t)ks * nsw + slice * NR) * 4 * 64 + r) * 8);
            *reinterpret_cast<u32x4*>(wl + (size_t)v * 8) = val;
        }
    }
    bf16* scr = wl + (size_t)p.ksteps * NR * 4 * 64 * 8 + wave * (32 * SCR_STR);
    __syncthreads();

    const int n = n0 + lr;                                  // + 32*nr
    const int ctot = p.C0 + p.C1;
    const int wstride = p.blocks_per_slice * 16;
    int tile = brank * 16 + wave;
    if (tile >= p.m_tiles) return;

    // ---- activation stream ---------------------------------------------------------------------------------------------
    bf16x8 a[D];
    int pf_tile = tile, pf_u = 0;                           // prefetch cursor
    auto a_issue = [&](bf16x8& dst) __attribute__((always_inline)) {
        const int t = pf_tile < p.m_tiles ? pf_tile : tile;
        int row = t * 32 + lr;
        row = row < p.M ? row : p.M - 1;
        int k = pf_u * 16 + 8 * lh;
        k = k < ctot ? k : 0;                               // zero-weight padding: any finite data will do
        const bf16* src = k < p.C0 ? p.src0 + (size_t)row * p.C0 + k : p.src1 + (size_t)row * p.C1 + (k - p.C0);
        dst = *reinterpret_cast<const bf16x8*>(src);
        if (++pf_u == U) { pf_u = 0; pf_tile += wstride; }
    };
#pragma unroll
    for (int i = 0; i < D; ++i) a_issue(a[i]);

    float bias_v[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) bias_v[nr] = (p.bias && n + 32 * nr < p.Cout) ? p.bias[n + 32 * nr] : 0.f;

    for (; tile < p.m_tiles; tile += wstride) {
        f32x16 acc[NR];
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[nr][j] = bias_v[nr];
        for (int u0 = 0; u0 < U; u0 += D) {                 // U is a multiple of D (K padded to 128)
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int u = u0 + i;
                const bf16* wb = wl + ((size_t)(u >> 2) * NR * 4 + (u & 3)) * 64 * 8 + lane * 8;
                bf16x8 bfr[NR];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bfr[nr] = *reinterpret_cast<const bf16x8*>(wb + (size_t)nr * 4 * 64 * 8);
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bfr[nr], acc[nr], 0, 0, 0);
                a_issue(a[i]);                              // refill this ring slot D k-slices ahead
            }
        }
        // ---- epilogue: transpose 32 px x 32 couts through the wave's LDS patch, 16-B vectors out ------------------------
        const int pix0 = tile * 32;
        const int b = pix0 / p.HW;
        const int pl0 = lane >> 2, cq = lane & 3;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
                scr[row * SCR_STR + lr] = (bf16)acc[nr][j];
            }
            float vs1[8], vs2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) vs1[k] = vs2[k] = 0.f;
            const int cbase = n0 + 32 * nr + cq * 8;
#pragma unroll
            for (int v2 = 0; v2 < 2; ++v2) {
                const int pl = pl0 + 16 * v2;
                const bf16x8 raw = *reinterpret_cast<const bf16x8*>(scr + pl * SCR_STR + cq * 8);
                const int pix = pix0 + pl;
                if (pix < p.M && cbase < p.Cout) {
                    const size_t off = (size_t)pix * p.Cout + cbase;
                    float f[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = (float)raw[k];
                    bf16x8 o = raw;
                    if (p.res) {
                        const bf16x8 rr = *reinterpret_cast<const bf16x8*>(p.res + off);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            o[k] = (bf16)fmaf(p.res_scale, f[k], (float)rr[k]);
                            f[k] = (float)o[k];
                        }
                    }
                    *reinterpret_cast<bf16x8*>(p.out + off) = o;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { vs1[k] += f[k]; vs2[k] = fmaf(f[k], f[k], vs2[k]); }
                }
            }
            if (p.stats) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
#pragma unroll
                    for (int o2 = 4; o2 < 64; o2 <<= 1) {
                        vs1[k] += __shfl_xor(vs1[k], o2, 64);
                        vs2[k] += __shfl_xor(vs2[k], o2, 64);
                    }
                }
                if (lane < 4 && cbase < p.Cout) {
                    const int tiles_per_img = p.HW / 32;
                    float2* dst = p.stats + ((size_t)b * tiles_per_img + (tile - b * tiles_per_img)) * p.Cout + cbase;
#pragma unroll
                    for (int k = 0; k < 8; ++k) dst[k] = make_float2(vs1[k], vs2[k]);
                }
            }
        }
    }
}

static int g_cus = 0;

template <int BN>
static int run_c1(C1Params& p, hipStream_t s) {
    const size_t lds = (size_t)p.ksteps * (BN / 32) * 4 * 64 * 16 + 16 * 32 * 40 * 2;
    if (lds > 160 * 1024) return HSIDM_E_UNSUPPORTED;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_kernel<BN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    if (g_cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g_cus <= 0)
            g_cus = 256;
    }
    p.n_slices = p.Cout_pad / BN;
    p.m_tiles = (p.M + 31) / 32;
    int bps = g_cus / p.n_slices;                               // workgroups per cout slice
    if (bps < 1) bps = 1;
    const int need = (p.m_tiles + 15) / 16;
    if (bps > need) bps = need;
    p.blocks_per_slice = bps;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_ws_kernel<BN>), dim3(bps * p.n_slices), dim3(1024), lds, s, p);
    return (int)hipGetLastError();
}

// K_pad = ksteps*64 <= 512 -> BN 64; <= 1024 -> BN 32
int conv1x1_ws_run(C1Params& p, hipStream_t s) {
    if (p.ksteps <= 8 && p.Cout_pad % 64 == 0) return run_c1<64>(p, s);
    if (p.ksteps <= 16) return run_c1<32>(p, s);
    return HSIDM_E_UNSUPPORTED;
}

}  // namespace hsidm
