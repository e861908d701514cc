// Rate of the 2:4 structured-sparse matrix instruction v_smfmac_f32_16x16x64_f16 against the dense v_mfma_f32_16x16x32_f16 in the loop of
// mfma_shape.hip (dense operand by ds_read_b128 from LDS, the other from an L2-resident array), logical FLOP counted for both.
//   hipcc --offload-arch=gfx950 -O3 -o smfmac_rate smfmac_rate.hip && ./smfmac_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 80, ROWS = 160, KSTEPS32 = 18;
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const f16* __restrict__ a_src, const f16* __restrict__ w, const int* __restrict__ idxs, float* __restrict__ out, int items) {
    __shared__ __attribute__((aligned(16))) f16 tile[ROWS * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ROWS * PITCH / 8; i += 256) reinterpret_cast<f16x8*>(tile)[i] = reinterpret_cast<const f16x8*>(a_src)[(i + blockIdx.x * 7) % (ROWS * PITCH / 8)];
    __syncthreads();
    float sum = 0.f;
    const int lr = lane & 15, lq = lane >> 4;
    const f16* ab = tile + lr * PITCH + 8 * lq;
    const f16* wl = w + ((size_t)wave * 64 + lane) * 8;
    const int idx = idxs[lane];
    for (int it = 0; it < items; ++it) {
        f32x4 acc[8][2];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) {
#pragma unroll 2
            for (int s = 0; s < KSTEPS32; ++s) {
                f16x8 b[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const f16x8*>(wl + (size_t)(s * 2 + n) * 4 * 64 * 8);
                const int off = (s % 2) * 32 + (s / 2 % 3) * PITCH;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const f16x8 a = *reinterpret_cast<const f16x8*>(ab + m * 16 * PITCH + off);
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n], a, acc[m][n], 0, 0, 0);
                }
            }
        } else {
#pragma unroll 1
            for (int s = 0; s < KSTEPS32 / 2; ++s) {        // one step = 64 k
                f16x8 b[2];                                  // compressed sparse operand: 8 stored values per lane
#pragma unroll
                for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const f16x8*>(wl + (size_t)(s * 2 + n) * 4 * 64 * 8);
                const int off = (s % 3) * PITCH;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const f16x8 a0 = *reinterpret_cast<const f16x8*>(ab + m * 16 * PITCH + off);
                    const f16x8 a1 = *reinterpret_cast<const f16x8*>(ab + m * 16 * PITCH + off + 32);
                    const f16x16 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(b[n], a, acc[m][n], idx, 0, 0);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += acc[m][n][j];
    }
    out[blockIdx.x * 256 + tid] = sum;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    const int G = 512, items = 200;
    std::vector<f16> ha(ROWS * PITCH), hw(36 * 4 * 64 * 8 * 2);
    std::vector<int> hi(64);
    srand(1);
    for (auto& v : ha) v = (f16)((rand() % 2001 - 1000) / 500.0f);
    for (auto& v : hw) v = (f16)((rand() % 2001 - 1000) / 5000.0f);
    for (auto& v : hi) v = 0x44444444;          // every pair of stored values: positions 0 and 1 of its group of four
    f16 *da, *dw; int* di; float* dout;
    CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&di, 256)); CK(hipMalloc(&dout, G * 256 * 4));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(di, hi.data(), 256, hipMemcpyHostToDevice));
    const double flop = 2.0 * G * 4 * 128 * 32 * 576 * items;
    auto run = [&](int mode, int n) {
        for (int i = 0; i < n; ++i) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, da, dw, di, dout, items);
            else hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, da, dw, di, dout, items);
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) { run(0, 10); run(1, 10); CK(hipDeviceSynchronize()); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round)
        for (int mode = 0; mode < 2; ++mode) {
            run(mode, 20);
            CK(hipEventRecord(e0)); run(mode, 100); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s: %.3f ms per launch, %.1f TFLOP/s (logical)\n", mode == 0 ? "dense 16x16x32" : "sparse 16x16x64", ms / 100, flop / (ms / 100 * 1e-3) / 1e12);
        }
    return 0;
}
