// Which bf16/fp16 MFMA shape sustains more FLOP/s under the chip's power management, in a loop shaped like the conv kernels' main loop?
// (MI355X_MICROARCH.md, "DVFS give-back" item 7: equal cycles per FLOP, different clocks.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
// One workgroup = 4 waves, two workgroups per CU; a wave owns 128 rows x 32 columns of output per item and walks K = 576 per item:
//   A fragments by ds_read_b128 from a 160-row LDS tile (144-byte rows, random data), B fragments from an L2-resident global array
//   (16 B per lane and fragment, prefetched one step ahead), f32 accumulators.
//   MODE 0: v_mfma_f32_32x32x16_f16   - per k = 16: 4 A + 1 B fragments, 4 MFMAs
//   MODE 1: v_mfma_f32_16x16x32_f16   - per k = 32: 8 A + 2 B fragments, 16 MFMAs        (same LDS and global bytes per FLOP)
// Prints wall time and TFLOP/s of each mode, alternating, after a 2 s warm-up of back-to-back launches.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PITCH = 72;          // f16 per LDS row
constexpr int ROWS = 160;
constexpr int KSTEPS16 = 36;       // K = 576
#ifndef SWZ
#define SWZ 1                      // k-group swizzle of tile columns 4..11: makes the 16x16x32 A reads conflict-free (0: 2-way)
#endif

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const f16* __restrict__ a_src, const f16* __restrict__ w, float* __restrict__ out, int items) {
    __shared__ __attribute__((aligned(16))) f16 tile[ROWS * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ROWS * PITCH / 8; i += 256)
        reinterpret_cast<f16x8*>(tile)[i] = reinterpret_cast<const f16x8*>(a_src)[(i + blockIdx.x * 7) % (ROWS * PITCH / 8)];
    __syncthreads();
    float sum = 0.f;
    if (MODE == 0) {
        const int lr = lane & 31, lh = lane >> 5;
        const f16* ab = tile + lr * PITCH + 8 * lh;
        const f16* wl = w + ((size_t)wave * 64 + lane) * 8;
        for (int it = 0; it < items; ++it) {
            f32x16 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;
            f16x8 b = *reinterpret_cast<const f16x8*>(wl);
#pragma unroll 4
            for (int s = 0; s < KSTEPS16; ++s) {
                const f16x8 bn = *reinterpret_cast<const f16x8*>(wl + (size_t)((s + 1) % KSTEPS16) * 4 * 64 * 8);
                const int off = (s % 4) * 16 + (s / 4 % 3) * PITCH;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f16x8 a = *reinterpret_cast<const f16x8*>(ab + m * 32 * PITCH + off);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
                }
                b = bn;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 16; ++j) sum += acc[m][j];
        }
    } else {
        const int lr = lane & 15, lq = lane >> 4;
        const f16* ab = tile + lr * PITCH + 8 * (lq ^ ((lr >= 4 && lr < 12) ? SWZ : 0));
        const f16* wl = w + ((size_t)wave * 64 + lane) * 8;
        for (int it = 0; it < items; ++it) {
            f32x4 acc[8][2];
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            f16x8 b[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const f16x8*>(wl + n * 4 * 64 * 8);
#pragma unroll 2
            for (int s = 0; s < KSTEPS16 / 2; ++s) {
                f16x8 bn[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) bn[n] = *reinterpret_cast<const f16x8*>(wl + (size_t)(((s + 1) % (KSTEPS16 / 2)) * 2 + n) * 4 * 64 * 8);
                const int off = (s % 2) * 32 + (s / 2 % 3) * PITCH;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const f16x8 a = *reinterpret_cast<const f16x8*>(ab + m * 16 * PITCH + off);
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[n], acc[m][n], 0, 0, 0);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) b[n] = bn[n];
            }
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum += acc[m][n][j];
        }
    }
    out[blockIdx.x * 256 + tid] = sum;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
    const int G = 512, items = 200;
    std::vector<f16> ha(ROWS * PITCH), hw(36 * 4 * 64 * 8 * 2);
    srand(1);
    for (auto& v : ha) v = (f16)((rand() % 2001 - 1000) / 500.0f);
    for (auto& v : hw) v = (f16)((rand() % 2001 - 1000) / 5000.0f);
    f16 *da, *dw;
    float* dout;
    CK(hipMalloc(&da, ha.size() * 2));
    CK(hipMalloc(&dw, hw.size() * 2));
    CK(hipMalloc(&dout, G * 256 * 4));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const double flop = 2.0 * G * 4 * 128 * 32 * 576 * items;
    auto run = [&](int mode, int n) {
        for (int i = 0; i < n; ++i) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, da, dw, dout, items);
            else hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, da, dw, dout, items);
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) { run(0, 10); run(1, 10); CK(hipDeviceSynchronize()); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round)
        for (int mode = 0; mode < 2; ++mode) {
            run(mode, 20);
            CK(hipEventRecord(e0));
            run(mode, 100);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s: %.3f ms per launch, %.1f TFLOP/s\n", mode == 0 ? "32x32x16" : "16x16x32", ms / 100, flop / (ms / 100 * 1e-3) / 1e12);
        }
    return 0;
}
