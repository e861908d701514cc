// Shared device helpers for the hsidm HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace hsidm {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int kWave = 64;

// ---- bf16 <-> f32 -------------------------------------------------------------------------
// plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-preserving) on gfx950
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }

// ---- the 16-bit element type of a throughput mode --------------------------------------------------
// bf16 (8-bit significand) or fp16 (11-bit significand, |x| <= 65504): same MFMA rate, same storage, same layouts; every
// 16-bit kernel is a template over it.  lo / hi: the two elements of a packed 32-bit word as fp32 (for fp16 the conversion
// folds into the consumer: v_fma_mix_f32 reads fp16 halves directly); sat: the clamp applied to values on their way to an
// fp16 STORE (an overflow would otherwise become inf and poison the rest of the chain); operands staged behind
// GroupNorm + SiLU are bounded and skip it.
template <typename E> struct Elem;
template <> struct Elem<bf16> {
    using x8 = bf16x8; using x4 = bf16x4; using x2 = bf16x2;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    // 16x16x32: lane (c = lane % 16, g = lane / 16) holds A[row c][k 8g..8g+7], B[k 8g..8g+7][col c]; C register j = [row 4g + j][col c]
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
    static __device__ __forceinline__ float sat(float v) { return v; }
};
template <> struct Elem<f16> {
    using x8 = f16x8; using x4 = f16x4; using x2 = f16x2;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }
    static __device__ __forceinline__ float sat(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
};
// The conv kernels saturate their fp16 stores in HARDWARE: with MODE.FP16_OVFL set, a conversion to fp16 that overflows gives
// +-65504 instead of an infinity (a true infinity stays one; probed on gfx950 for v_cvt_pk_f16_f32 and v_cvt_f16_f32).  A wave that
// called fp16_saturating_stores() converts with cvt_pair_hw / a plain cast - without the v_med3 of Elem<f16>::sat, 64 ... 128 vector
// instructions per lane and work item in the convolution epilogues.  (hwreg(MODE, offset 23, size 1) = 1473.)
// CONTRACT: cvt_pair_hw / a plain (f16) cast saturate ONLY in a kernel that called fp16_saturating_stores() at entry (the compiler does not
// model the mode bit; every fp16 kernel with such conversions does so in its first statements, and tests/test_gpu_anchor.py::
// test_fp16_stores_saturate_instead_of_overflowing drives every store path of every one of them past 65 504).  A kernel without the call
// must convert with cvt_pair (v_med3 clamp).  Difference to the clamp: a TRUE +-inf or NaN accumulator passes through unchanged here (the
// clamp would turn +-inf into +-65 504) - such a value can only come from non-finite inputs, which the path does not produce.
__device__ __forceinline__ void fp16_saturating_stores() { __builtin_amdgcn_s_setreg(1473, 1); }
template <typename E>
__device__ __forceinline__ typename Elem<E>::x2 cvt_pair_hw(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_convertvector(v, typename Elem<E>::x2);
}
// two fp32 -> one packed pair (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round to nearest even), saturated for fp16
template <typename E>
__device__ __forceinline__ typename Elem<E>::x2 cvt_pair(float a, float b) {
    const f32x2v v = {Elem<E>::sat(a), Elem<E>::sat(b)};
    return __builtin_convertvector(v, typename Elem<E>::x2);
}

// 8 consecutive activations as fp32, from either storage type
template <typename T> struct Vec8;
template <> struct Vec8<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) {
        bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) {
        bf16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (bf16)v[i];
        *reinterpret_cast<bf16x8*>(p) = r;
    }
};
template <> struct Vec8<f16> {
    static __device__ __forceinline__ void load(const f16* p, float (&v)[8]) {
        f16x8 r = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    }
    static __device__ __forceinline__ void store(f16* p, const float (&v)[8]) {
        f16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (f16)Elem<f16>::sat(v[i]);
        *reinterpret_cast<f16x8*>(p) = r;
    }
};
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        f32x4 a = *reinterpret_cast<const f32x4*>(p);
        f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        f32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
};

template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 x) { return (float)x; }
template <> __device__ __forceinline__ float to_f32<f16>(f16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float x) { return (f16)Elem<f16>::sat(x); }

// v_exp_f32 + v_rcp_f32 (~1e-6 relative), 5 VALU instead of an IEEE division
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---- wave / block reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- host-side helpers ------------------------------------------------------------------------
// One-time set-up per (call site, DEVICE): function attributes such as the dynamic-LDS cap belong to the device the
// calling thread has current, so a process that drives several devices (one thread each, like nn.DataParallel) must
// repeat them per device.  Lock-free; the guarded calls are idempotent, so a race only repeats one.
struct PerDeviceOnce { std::atomic<uint64_t> mask{0}; };
template <typename F>
inline int per_device_once(PerDeviceOnce& st, F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    if (st.mask.load(std::memory_order_acquire) & bit) return 0;
    const int rc = f();
    if (rc) return rc;
    st.mask.fetch_or(bit, std::memory_order_release);
    return 0;
}
template <typename K>
inline int raise_lds_cap(PerDeviceOnce& st, K kernel, size_t bytes) {
    return per_device_once(st, [&] { return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); });
}
// compute units of the calling thread's current device (cached per device)
inline int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int c = cache[dev & 63].load(std::memory_order_relaxed);
    if (c > 0) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cache[dev & 63].store(c, std::memory_order_relaxed);
    return c;
}
// Diagnostic switches (A/B measurements and tests): set once from the environment when the library is loaded
// (HSIDM_NO_V3, HSIDM_V2_BN256, HSIDM_ATTENTION_V1, HSIDM_NO_XCD_MAP, HSIDM_1X1, HSIDM_V2_ABL, HSIDM_NO_SPLIT_K) and afterwards only through
// hsidm_debug_switch(); the launch path never reads the environment.
enum DebugKey { DBG_NO_V3 = 0, DBG_V2_BN256, DBG_ATTENTION_V1, DBG_NO_XCD_MAP, DBG_1X1_V1, DBG_V2_ABL, DBG_SK_MULT, DBG_NO_SPLIT_K, DBG_NO_SPARSE_LO, DBG_NO_FUSED_PROJ, DBG_COUNT };
int debug_get(int key);

}  // namespace hsidm
