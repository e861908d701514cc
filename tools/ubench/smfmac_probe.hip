// Operand semantics of v_smfmac_f32_16x16x64_f16 (2:4 structured-sparse A), found by experiment - the ISA text is not in this image.
//   hipcc --offload-arch=gfx950 -O2 -o smfmac_probe smfmac_probe.hip && ./smfmac_probe > smfmac_probe.txt
// Per lane: A = 8 stored halfs (f16x8), B = 16 halfs (f16x16), idx = 32-bit index register, D = 4 floats.
// Experiment 1: A one-hot at stored slot sa of lanes with k-group ga (every row), the slot pair's 2-bit index fields = (v0, v1), B one-hot at
//   (k-group gb, slot sb) of every column -> D == 1 everywhere iff the hardware pairs that stored value with that B element.
// Experiment 2: A one-hot in ONE row, B all ones -> which (lane, register) of D is that row.
// Experiment 3: which 16 bits of idx are read for abid = 0 / 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// block = one configuration (ga, sa, v0, v1); loops over the 64 B positions; out[cfg][bpos] = sum of all D entries
template <int ABID>
__global__ void exp1(float* out) {
    const int cfg = blockIdx.x;
    const int v1 = cfg & 3, v0 = (cfg >> 2) & 3, sa = (cfg >> 4) & 7, ga = cfg >> 7;
    const int lane = threadIdx.x, g = lane >> 4;
    f16x8 a;
    for (int s = 0; s < 8; ++s) a[s] = (f16)((g == ga && s == sa) ? 1.0f : 0.0f);
    // index fields: slot s uses bits [2s+1 : 2s]; the pair of sa gets (v0, v1), every other pair (0, 1)
    unsigned idx16 = 0;
    for (int m = 0; m < 4; ++m) {
        const int f0 = (m == sa / 2) ? v0 : 0, f1 = (m == sa / 2) ? v1 : 1;
        idx16 |= (unsigned)(f0 | (f1 << 2)) << (4 * m);
    }
    const int idx = ABID == 0 ? (int)(idx16 | 0xe4e40000u) : (int)((idx16 << 16) | 0xe4e4u);   // the unused half holds another valid pattern
    for (int bp = 0; bp < 64; ++bp) {
        const int gb = bp >> 4, sb = bp & 15;
        f16x16 b;
        for (int s = 0; s < 16; ++s) b[s] = (f16)((g == gb && s == sb) ? 1.0f : 0.0f);
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, idx, 0, ABID);
        float sum = c[0] + c[1] + c[2] + c[3];
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) out[cfg * 64 + bp] = sum;
    }
}

__global__ void exp2(float* out) {      // block = row i0: A one-hot (slot 0 of k-group 0) in lane i0 only, B all ones
    const int i0 = blockIdx.x, lane = threadIdx.x;
    f16x8 a;
    for (int s = 0; s < 8; ++s) a[s] = (f16)((lane == i0 && s == 0) ? 1.0f : 0.0f);
    f16x16 b;
    for (int s = 0; s < 16; ++s) b[s] = (f16)1.0f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, 0x44444444, 0, 0);
    for (int r = 0; r < 4; ++r) out[(i0 * 64 + lane) * 4 + r] = c[r];
}

__global__ void exp2b(float* out) {     // block = column j0: A all ones, B one-hot (k-group 0 slot 0) in lane j0 only
    const int j0 = blockIdx.x, lane = threadIdx.x;
    f16x8 a;
    for (int s = 0; s < 8; ++s) a[s] = (f16)1.0f;
    f16x16 b;
    for (int s = 0; s < 16; ++s) b[s] = (f16)((lane == j0 && s == 0) ? 1.0f : 0.0f);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, 0x44444444, 0, 0);
    for (int r = 0; r < 4; ++r) out[(j0 * 64 + lane) * 4 + r] = c[r];
}

// a full random check of the model the first experiments suggest is done on the host side of the product's tests, not here

int main() {
    float *d1, *d1b, *d2, *d2b;
    const int n1 = 512 * 64, n2 = 16 * 64 * 4;
    CK(hipMalloc(&d1, n1 * 4)); CK(hipMalloc(&d1b, n1 * 4)); CK(hipMalloc(&d2, n2 * 4)); CK(hipMalloc(&d2b, n2 * 4));
    hipLaunchKernelGGL(exp1<0>, dim3(512), dim3(64), 0, 0, d1);
    hipLaunchKernelGGL(exp1<1>, dim3(512), dim3(64), 0, 0, d1b);
    hipLaunchKernelGGL(exp2, dim3(16), dim3(64), 0, 0, d2);
    hipLaunchKernelGGL(exp2b, dim3(16), dim3(64), 0, 0, d2b);
    CK(hipDeviceSynchronize());
    std::vector<float> h1(n1), h1b(n1), h2(n2), h2b(n2);
    CK(hipMemcpy(h1.data(), d1, n1 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1b.data(), d1b, n1 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), d2, n2 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2b.data(), d2b, n2 * 4, hipMemcpyDeviceToHost));
    for (int ab = 0; ab < 2; ++ab) {
        const std::vector<float>& h = ab ? h1b : h1;
        for (int cfg = 0; cfg < 512; ++cfg) {
            const int v1 = cfg & 3, v0 = (cfg >> 2) & 3, sa = (cfg >> 4) & 7, ga = cfg >> 7;
            printf("exp1 abid=%d ga=%d sa=%d v0=%d v1=%d :", ab, ga, sa, v0, v1);
            for (int bp = 0; bp < 64; ++bp)
                if (h[cfg * 64 + bp] != 0.f) printf(" (gb=%d sb=%d sum=%g)", bp >> 4, bp & 15, h[cfg * 64 + bp]);
            printf("\n");
        }
    }
    for (int i0 = 0; i0 < 16; ++i0) {
        printf("exp2 row-onehot lane=%d :", i0);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h2[(i0 * 64 + l) * 4 + r] != 0.f) printf(" (lane=%d r=%d %g)", l, r, h2[(i0 * 64 + l) * 4 + r]);
        printf("\n");
    }
    for (int j0 = 0; j0 < 16; ++j0) {
        printf("exp2b col-onehot lane=%d :", j0);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h2b[(j0 * 64 + l) * 4 + r] != 0.f) printf(" (lane=%d r=%d %g)", l, r, h2b[(j0 * 64 + l) * 4 + r]);
        printf("\n");
    }
    return 0;
}
