// Patch preparation on the device: the step right before GAE.encode in the reference's data path.
//   resample_axis     : one axis of the MATLAB-compatible bicubic resize (imsize.py:35-158): out[o][j][i] =
//                       sum_p w[j][p] * src[o][idx[j][p]][i]; the tap tables (antialiased kernel, mirrored edges) are built on
//                       the host once per (in, out) length pair; HStest.py:40-45 calls it as x1/4 then x4.
//   minmax_normalize  : (img - min) / (max - min) over a whole cube (HStest.py:37, HStrain.py:49), deterministic
//                       two-stage min/max.
//   augment           : the 8 flips / quarter turns of the training set (utils.py:3-28, caller HStrain.py:65) as one gather
//                       over the two spatial axes of an NCHW batch.
//   color_correction  : per-band mean / standard-deviation matching to a guide cube, clipped to [0, 1]
//                       (eval_hsi.py:259-274, caller sr_gae.py:340); band moments in fp64, fixed reduction order.
// Streaming kernels, HBM/latency bound; fp32 accumulation (the reference accumulates in float64 on the host).
#include "common.h"
#include "../../include/hsidm.h"

namespace hsidm {

__global__ __launch_bounds__(256) void resample_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, int in_len, int out_len,
                                                            int inner, const float* __restrict__ w, const int32_t* __restrict__ idx,
                                                            int taps, int clamp01, int64_t total) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int i = (int)(g % inner);
        const int64_t r = g / inner;
        const int j = (int)(r % out_len);
        const int64_t o = r / out_len;
        const float* s = src + o * in_len * (int64_t)inner + i;
        float acc = 0.f;
        for (int p = 0; p < taps; ++p) acc = fmaf(w[j * taps + p], s[(int64_t)idx[j * taps + p] * inner], acc);
        if (clamp01) acc = fminf(1.f, fmaxf(0.f, acc));
        dst[g] = acc;
    }
}

constexpr int kMinMaxBlocks = 64;

__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ ws) {
    __shared__ float smin[256], smax[256];
    const int p = blockIdx.y;
    const float* c = x + (int64_t)p * n;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kMinMaxBlocks * 256) {
        const float v = c[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ws[((int64_t)p * kMinMaxBlocks + blockIdx.x) * 2] = smin[0];
        ws[((int64_t)p * kMinMaxBlocks + blockIdx.x) * 2 + 1] = smax[0];
    }
}

__global__ __launch_bounds__(256) void minmax_apply_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                                           const float* __restrict__ ws) {
    const int p = blockIdx.y;
    float lo = INFINITY, hi = -INFINITY;
    for (int b = 0; b < kMinMaxBlocks; ++b) {                 // 64 partials: every thread folds them itself (fixed order)
        lo = fminf(lo, ws[((int64_t)p * kMinMaxBlocks + b) * 2]);
        hi = fmaxf(hi, ws[((int64_t)p * kMinMaxBlocks + b) * 2 + 1]);
    }
    const float d = hi - lo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[(int64_t)p * n + i] = (x[(int64_t)p * n + i] - lo) / d;
}

// dst[o][i][j] = src[o][si][sj]; src planes are H x W, dst planes H x W (modes 0, 1, 4, 5) or W x H (2, 3, 6, 7).
// numpy semantics: flipud reverses axis 0, rot90 turns counter-clockwise in the (0, 1) plane (utils.py:3-28):
//   0 identity            (i, j)            4 rot180            (H-1-i, W-1-j)
//   1 flipud              (H-1-i, j)        5 flipud(rot180)    (i, W-1-j)
//   2 rot90               (j, W-1-i)        6 rot270            (H-1-j, i)
//   3 flipud(rot90)       (j, i)            7 flipud(rot270)    (H-1-j, W-1-i)
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int mode,
                                                      int64_t total) {
    const bool turn = (mode == 2) | (mode == 3) | (mode == 6) | (mode == 7);
    const int OH = turn ? W : H, OW = turn ? H : W;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int j = (int)(g % OW);
        const int64_t r = g / OW;
        const int i = (int)(r % OH);
        const int64_t o = r / OH;
        int si, sj;
        switch (mode) {
            case 0: si = i; sj = j; break;
            case 1: si = H - 1 - i; sj = j; break;
            case 2: si = j; sj = W - 1 - i; break;
            case 3: si = j; sj = i; break;
            case 4: si = H - 1 - i; sj = W - 1 - j; break;
            case 5: si = i; sj = W - 1 - j; break;
            case 6: si = H - 1 - j; sj = i; break;
            default: si = H - 1 - j; sj = W - 1 - i; break;
        }
        dst[g] = src[(o * H + si) * W + sj];
    }
}

// moments[(p*C + c)*2 + {0,1}] = {sum x, sum x^2} of one band, fp64 throughout (var = E[x^2] - mean^2 cancels badly in fp32)
__global__ __launch_bounds__(256) void band_moments_kernel(const float* __restrict__ x, int HW, double* __restrict__ moments) {
    __shared__ double r0[256], r1[256];
    const float* b = x + (size_t)blockIdx.x * HW;
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const double v = (double)b[i];
        s0 += v;
        s1 += v * v;
    }
    r0[threadIdx.x] = s0;
    r1[threadIdx.x] = s1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            r0[threadIdx.x] += r0[threadIdx.x + s];
            r1[threadIdx.x] += r1[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        moments[(size_t)blockIdx.x * 2] = r0[0];
        moments[(size_t)blockIdx.x * 2 + 1] = r1[0];
    }
}

// out = clip((x - mean_x) / std_x * std_g + mean_g, 0, 1) for bands < nch, 0 for the others (eval_hsi.py:268-274:
// the output starts as zeros and only the first num_channels bands are filled)
__global__ __launch_bounds__(256) void color_apply_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int HW, int guide_HW,
                                                          int nch, const double* __restrict__ mg, const double* __restrict__ mx) {
    const int bc = blockIdx.y;                                // p*C + c
    const int c = bc % C;
    if (c < nch) {
        const double gm = mg[bc * 2] / guide_HW, xm = mx[bc * 2] / HW;
        const double gs = sqrt(fmax(mg[bc * 2 + 1] / guide_HW - gm * gm, 0.0)), xs = sqrt(fmax(mx[bc * 2 + 1] / HW - xm * xm, 0.0));
        // the reference works in the arrays' float32: (x - m) / s * gs + gm, evaluated left to right
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            const float v = ((x[(size_t)bc * HW + i] - (float)xm) / (float)xs) * (float)gs + (float)gm;
            out[(size_t)bc * HW + i] = fminf(1.f, fmaxf(0.f, v));
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) out[(size_t)bc * HW + i] = 0.f;
}

}  // namespace hsidm

extern "C" int hsidm_resample_axis(const float* src, float* dst, int64_t outer, int in_len, int out_len, int inner, const float* weights,
                                   const int32_t* indices, int taps, int clamp01, void* stream) {
    if (!src || !dst || !weights || !indices || outer <= 0 || in_len <= 0 || out_len <= 0 || inner <= 0 || taps <= 0) return HSIDM_E_BADARG;
    const int64_t total = outer * out_len * inner;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(hsidm::resample_axis_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, dst, in_len, out_len, inner,
                       weights, indices, taps, clamp01, total);
    return (int)hipGetLastError();
}

extern "C" int hsidm_minmax_workspace_bytes(int P) { return P <= 0 ? HSIDM_E_BADARG : P * hsidm::kMinMaxBlocks * 2 * (int)sizeof(float); }

extern "C" int hsidm_minmax_normalize(const float* x, float* out, int P, int64_t n, void* workspace, void* stream) {
    if (!x || !out || !workspace || P <= 0 || n <= 0) return HSIDM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(hsidm::minmax_partial_kernel, dim3(hsidm::kMinMaxBlocks, P), dim3(256), 0, s, x, n, (float*)workspace);
    int64_t g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(hsidm::minmax_apply_kernel, dim3((unsigned)g, P), dim3(256), 0, s, x, out, n, (const float*)workspace);
    return (int)hipGetLastError();
}

extern "C" int hsidm_augment(const float* src, float* dst, int64_t outer, int H, int W, int mode, void* stream) {
    if (!src || !dst || src == dst || outer <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 7) return HSIDM_E_BADARG;
    const int64_t total = outer * H * W;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(hsidm::augment_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, mode, total);
    return (int)hipGetLastError();
}

extern "C" int hsidm_color_correction_workspace_bytes(int P, int C) {
    return (P <= 0 || C <= 0) ? HSIDM_E_BADARG : P * C * 4 * (int)sizeof(double);
}

extern "C" int hsidm_color_correction(const float* guide, int guide_HW, const float* x, float* out, int P, int C, int HW, int num_channels,
                                      void* workspace, void* stream) {
    if (!guide || !x || !out || !workspace || P <= 0 || C <= 0 || HW <= 0 || guide_HW <= 0 || num_channels < 0) return HSIDM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    double* mg = (double*)workspace;
    double* mx = mg + (size_t)P * C * 2;
    hipLaunchKernelGGL(hsidm::band_moments_kernel, dim3(P * C), dim3(256), 0, s, guide, guide_HW, mg);
    hipLaunchKernelGGL(hsidm::band_moments_kernel, dim3(P * C), dim3(256), 0, s, x, HW, mx);
    int g = (HW + 255) / 256;
    if (g > 64) g = 64;
    hipLaunchKernelGGL(hsidm::color_apply_kernel, dim3(g, P * C), dim3(256), 0, s, x, out, C, HW, guide_HW, num_channels < C ? num_channels : C,
                       (const double*)mg, (const double*)mx);
    return (int)hipGetLastError();
}
