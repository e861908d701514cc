// Quality indices of a batch of hyperspectral cubes on the device: MPSNR, SAM, ERGAS, CC, RMSE of
// reference eval_hsi.py:27-121, the step right after GAE.decode in sr_gae.py:468-474 (today numpy on the host, SAM as a
// Python double loop over pixels).  Cubes are NCHW fp32 [P][C][H*W], as hsidm's decode returns them.
//   band_stats : grid (C, P)            six sums per band (t, p, t^2, p^2, t*p, (t-p)^2), fp64 across the block
//   sam_partial: grid (ceil(HW/256), P) spectral angle of one pixel per thread (bands strided by HW: coalesced), block sums
//   finalize   : grid (P)               folds both in a fixed order -> out[p] = {mpsnr, sam_deg, ergas, cc, rmse}
// Deterministic (no atomics).  One deviation from the reference: the cosine is clamped to [-1, 1] before acos (float32
// rounding can push identical spectra to 1 + 1e-7, where numpy returns NaN).
#include "common.h"
#include "../../include/hsidm.h"

namespace hsidm {

__device__ __forceinline__ double block_sum_f64(double v, double* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}

__global__ __launch_bounds__(256) void band_stats_kernel(const float* __restrict__ truth, const float* __restrict__ pred, int C, int HW,
                                                         double* __restrict__ stats) {
    __shared__ double red[256];
    const int c = blockIdx.x, p = blockIdx.y;
    const float* t = truth + ((size_t)p * C + c) * HW;
    const float* q = pred + ((size_t)p * C + c) * HW;
    // fp64 partials: CC is formed as E[tp] - E[t]E[p] over sqrt of two such differences, which for low-contrast bands
    // (variance << mean^2) needs more digits than fp32 sums of ~HW/256 terms keep; the kernel is HBM-bound either way
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < HW; i += 256) {
        const double a = t[i], b = q[i], d = a - b;
        s[0] += a; s[1] += b; s[2] += a * a; s[3] += b * b; s[4] += a * b; s[5] += d * d;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double tot = block_sum_f64(s[k], red);
        if (threadIdx.x == 0) stats[((size_t)p * C + c) * 6 + k] = tot;
    }
}

__global__ __launch_bounds__(256) void sam_partial_kernel(const float* __restrict__ truth, const float* __restrict__ pred, int C, int HW,
                                                          double* __restrict__ part) {
    __shared__ double red[256];
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    float ang = 0.f, cnt = 0.f;
    if (i < HW) {
        const float* t = truth + (size_t)p * C * HW + i;
        const float* q = pred + (size_t)p * C * HW + i;
        float dot = 0.f, nt = 0.f, nq = 0.f;
        for (int c = 0; c < C; ++c) {
            const float a = t[(size_t)c * HW], b = q[(size_t)c * HW];
            dot = fmaf(a, b, dot); nt = fmaf(a, a, nt); nq = fmaf(b, b, nq);
        }
        if (nt != 0.f && nq != 0.f) {                        // eval_hsi.py:60: pixels with a zero spectrum are skipped
            float cs = dot / (sqrtf(nt) * sqrtf(nq));
            cs = fminf(1.f, fmaxf(-1.f, cs));
            ang = acosf(cs);
            cnt = 1.f;
        }
    }
    const double sa = block_sum_f64((double)ang, red);
    const double sc = block_sum_f64((double)cnt, red);
    if (threadIdx.x == 0) {
        part[((size_t)p * gridDim.x + blockIdx.x) * 2] = sa;
        part[((size_t)p * gridDim.x + blockIdx.x) * 2 + 1] = sc;
    }
}

__global__ __launch_bounds__(256) void metrics_finalize_kernel(const double* __restrict__ stats, const double* __restrict__ part, int nblk, int C,
                                                               int HW, float ratio, float data_range, float* __restrict__ out) {
    __shared__ double red[256];
    const int p = blockIdx.x;
    double psnr = 0.0, erg = 0.0, cc = 0.0, sse = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        const double* s = stats + ((size_t)p * C + c) * 6;
        const double n = (double)HW;
        const double mse = s[5] / n, mt = s[0] / n;
        psnr += 10.0 * log10((double)data_range * data_range / mse);
        erg += mse / (mt * mt);
        cc += (s[4] - s[0] * s[1] / n) / sqrt((s[2] - s[0] * s[0] / n) * (s[3] - s[1] * s[1] / n));
        sse += s[5];
    }
    double ang = 0.0, cnt = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        ang += part[((size_t)p * nblk + b) * 2];
        cnt += part[((size_t)p * nblk + b) * 2 + 1];
    }
    psnr = block_sum_f64(psnr, red);
    erg = block_sum_f64(erg, red);
    cc = block_sum_f64(cc, red);
    sse = block_sum_f64(sse, red);
    ang = block_sum_f64(ang, red);
    cnt = block_sum_f64(cnt, red);
    if (threadIdx.x == 0) {
        float* o = out + (size_t)p * 5;
        o[0] = (float)(psnr / C);
        o[1] = (float)(ang / cnt * 180.0 / 3.14159265358979323846);
        o[2] = (float)((100.0 / ratio) * sqrt(erg / C));
        o[3] = (float)(cc / C);
        o[4] = (float)sqrt(sse / ((double)C * HW));
    }
}

// ---- MSSIM (eval_hsi.py:124-135: mean over bands of skimage.metrics.structural_similarity with its defaults: 7x7 uniform
// window, sample covariance (n/(n-1)), K1 = 0.01, K2 = 0.03, the 3-pixel border cropped before the mean) ----------------------
// grid (ceil((W-6)/16), ceil((H-6)/16), P*C), block 16x16: one interior pixel per thread, the 49 taps straight from L1/L2 (the whole
// cube is a few MB); fp64 block sums -> part[(p*C + c) * nblk + block]
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ truth, const float* __restrict__ pred, int H, int W,
                                                           float c1, float c2, double* __restrict__ part) {
    __shared__ double red[256];
    const int pc = blockIdx.z;
    const float* t = truth + (size_t)pc * H * W;
    const float* q = pred + (size_t)pc * H * W;
    const int x = 3 + blockIdx.x * 16 + (threadIdx.x & 15), y = 3 + blockIdx.y * 16 + (threadIdx.x >> 4);
    double sv = 0.0;
    if (x < W - 3 && y < H - 3) {
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
        for (int dy = -3; dy <= 3; ++dy)
#pragma unroll
            for (int dx = -3; dx <= 3; ++dx) {
                const float a = t[(size_t)(y + dy) * W + x + dx], b = q[(size_t)(y + dy) * W + x + dx];
                sx += a; sy += b; sxx = fmaf(a, a, sxx); syy = fmaf(b, b, syy); sxy = fmaf(a, b, sxy);
            }
        const float inv = 1.0f / 49.0f, cov = 49.0f / 48.0f;
        const float ux = sx * inv, uy = sy * inv;
        const float vx = cov * (sxx * inv - ux * ux), vy = cov * (syy * inv - uy * uy), vxy = cov * (sxy * inv - ux * uy);
        sv = (double)(((2.f * ux * uy + c1) * (2.f * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2)));
    }
    const double tot = block_sum_f64(sv, red);
    if (threadIdx.x == 0) part[((size_t)pc * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void ssim_finalize_kernel(const double* __restrict__ part, int C, int nblk, int H, int W, float* __restrict__ out) {
    __shared__ double red[256];
    const int p = blockIdx.x;
    double a = 0.0;
    for (int i = threadIdx.x; i < C * nblk; i += 256) a += part[(size_t)p * C * nblk + i];
    a = block_sum_f64(a, red);
    if (threadIdx.x == 0) out[p] = (float)(a / ((double)C * (H - 6) * (W - 6)));
}

}  // namespace hsidm

extern "C" int hsidm_hsi_mssim_workspace_bytes(int P, int C, int H, int W) {
    if (P <= 0 || C <= 0 || H < 7 || W < 7) return HSIDM_E_BADARG;
    const long long n = (long long)P * C * ((H - 6 + 15) / 16) * ((W - 6 + 15) / 16) * 8;
    return n > 0x7fffffffLL ? HSIDM_E_UNSUPPORTED : (int)n;
}

extern "C" int hsidm_hsi_mssim(const float* truth, const float* pred, int P, int C, int H, int W, float data_range, void* workspace,
                               float* out, void* stream) {
    if (!truth || !pred || !workspace || !out || P <= 0 || C <= 0 || H < 7 || W < 7 || !(data_range > 0.f) || (long long)P * C > 65535) return HSIDM_E_BADARG;
    const int gx = (W - 6 + 15) / 16, gy = (H - 6 + 15) / 16;
    const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(hsidm::ssim_partial_kernel, dim3(gx, gy, P * C), dim3(256), 0, s, truth, pred, H, W, c1, c2, (double*)workspace);
    hipLaunchKernelGGL(hsidm::ssim_finalize_kernel, dim3(P), dim3(256), 0, s, (const double*)workspace, C, gx * gy, H, W, out);
    return (int)hipGetLastError();
}

extern "C" int hsidm_hsi_metrics_workspace_bytes(int P, int C, int HW) {
    if (P <= 0 || C <= 0 || HW <= 0) return HSIDM_E_BADARG;
    const long long n = (long long)P * ((long long)C * 6 + (long long)((HW + 255) / 256) * 2) * 8;
    return n > 0x7fffffffLL ? HSIDM_E_UNSUPPORTED : (int)n;
}

extern "C" int hsidm_hsi_metrics(const float* truth, const float* pred, int P, int C, int HW, float ratio, float data_range,
                                 void* workspace, float* out, void* stream) {
    if (!truth || !pred || !workspace || !out || P <= 0 || C <= 0 || HW <= 0 || !(ratio > 0.f) || !(data_range > 0.f)) return HSIDM_E_BADARG;
    const int nblk = (HW + 255) / 256;
    double* stats = reinterpret_cast<double*>(workspace);
    double* part = stats + (size_t)P * C * 6;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(hsidm::band_stats_kernel, dim3(C, P), dim3(256), 0, s, truth, pred, C, HW, stats);
    hipLaunchKernelGGL(hsidm::sam_partial_kernel, dim3(nblk, P), dim3(256), 0, s, truth, pred, C, HW, part);
    hipLaunchKernelGGL(hsidm::metrics_finalize_kernel, dim3(P), dim3(256), 0, s, stats, part, nblk, C, HW, ratio, data_range, out);
    return (int)hipGetLastError();
}
