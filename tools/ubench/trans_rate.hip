// Issue cost of transcendental / packed-f16 VALU instructions on gfx950: one wave per SIMD runs N independent copies of one
// instruction in a loop; cycles per instruction = s_memtime delta / count.   hipcc --offload-arch=gfx950 -O2 -w trans_rate.hip
// Measured (round 1): v_add_f32 / v_and / v_lshlrev / v_fma_mix_f32 / v_cvt_pk_bf16_f32 / v_pk_*_f16 4 cycles; v_exp / v_rcp in
// f32 AND f16 8; v_fma_mixlo_f16 8; v_pk_add_f32 / v_pk_fma_f32 5 (two elements).  So an f16 SiLU pipeline buys nothing
// (the transcendentals are not cheaper and the f16 mixed FMA costs double).  v_cndmask_b32 with VCC reads 19 in this
// harness but 4 with an SGPR pair; replacing the staging commit's VCC selects by a mask and v_and made the real kernels
// 0.3 % SLOWER (one more instruction), so the 19 is an artefact of back-to-back VCC readers, not a price the kernels pay.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(ASM)                                                                                     \
    for (int it = 0; it < iters; ++it) {                                                              \
        asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); \
    }

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x * 0.001f + 0.5f, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f, a4 = a0 + 0.4f, a5 = a0 + 0.5f, a6 = a0 + 0.6f, a7 = a0 + 0.7f;
    unsigned long long t0, t1;
    asm volatile("s_mov_b64 vcc, exec\n s_mov_b64 s[10:11], exec" ::: "vcc", "s10", "s11");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (MODE == 0) BODY("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n")
    if (MODE == 1) BODY("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
    if (MODE == 2) BODY("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n")
    if (MODE == 3) BODY("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3\n v_rcp_f16 %4, %4\n v_rcp_f16 %5, %5\n v_rcp_f16 %6, %6\n v_rcp_f16 %7, %7\n")
    if (MODE == 4) BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
    if (MODE == 5) BODY("v_pk_add_f16 %0, %0, %0\n v_pk_add_f16 %1, %1, %1\n v_pk_add_f16 %2, %2, %2\n v_pk_add_f16 %3, %3, %3\n v_pk_add_f16 %4, %4, %4\n v_pk_add_f16 %5, %5, %5\n v_pk_add_f16 %6, %6, %6\n v_pk_add_f16 %7, %7, %7\n")
    if (MODE == 6) BODY("v_pk_mul_f16 %0, %0, %1\n v_pk_mul_f16 %1, %1, %2\n v_pk_mul_f16 %2, %2, %3\n v_pk_mul_f16 %3, %3, %4\n v_pk_mul_f16 %4, %4, %5\n v_pk_mul_f16 %5, %5, %6\n v_pk_mul_f16 %6, %6, %7\n v_pk_mul_f16 %7, %7, %0\n")
    if (MODE == 7) BODY("v_fma_mixlo_f16 %0, %0, %1, %2\n v_fma_mixlo_f16 %1, %1, %2, %3\n v_fma_mixlo_f16 %2, %2, %3, %4\n v_fma_mixlo_f16 %3, %3, %4, %5\n v_fma_mixlo_f16 %4, %4, %5, %6\n v_fma_mixlo_f16 %5, %5, %6, %7\n v_fma_mixlo_f16 %6, %6, %7, %0\n v_fma_mixlo_f16 %7, %7, %0, %1\n")
    if (MODE == 8) BODY("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n")
    if (MODE == 9) BODY("v_fma_mix_f32 %0, %0, %1, %2\n v_fma_mix_f32 %1, %1, %2, %3\n v_fma_mix_f32 %2, %2, %3, %4\n v_fma_mix_f32 %3, %3, %4, %5\n v_fma_mix_f32 %4, %4, %5, %6\n v_fma_mix_f32 %5, %5, %6, %7\n v_fma_mix_f32 %6, %6, %7, %0\n v_fma_mix_f32 %7, %7, %0, %1\n")
    if (MODE == 10) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
        for (int it = 0; it < iters; ++it)
            asm volatile(REP8(REP8("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        a0 = p0[0] + p1[0] + p2[0] + p3[1];
    }
    if (MODE == 11) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
        for (int it = 0; it < iters; ++it)
            asm volatile(REP8(REP8("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        a0 = p0[0] + p1[0] + p2[0] + p3[1];
    }
    if (MODE == 12) BODY("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n")
    if (MODE == 14) BODY("v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n v_cndmask_b32_e64 %4, %4, %5, s[10:11]\n v_cndmask_b32_e64 %5, %5, %6, s[10:11]\n v_cndmask_b32_e64 %6, %6, %7, s[10:11]\n v_cndmask_b32_e64 %7, %7, %0, s[10:11]\n")
    if (MODE == 15) BODY("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0\n")
    if (MODE == 16) BODY("v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %1, 0, %1, vcc\n v_cndmask_b32 %2, 0, %2, vcc\n v_cndmask_b32 %3, 0, %3, vcc\n v_cndmask_b32 %4, 0, %4, vcc\n v_cndmask_b32 %5, 0, %5, vcc\n v_cndmask_b32 %6, 0, %6, vcc\n v_cndmask_b32 %7, 0, %7, vcc\n")
    if (MODE == 17) BODY("v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n v_lshlrev_b32 %4, 16, %4\n v_lshlrev_b32 %5, 16, %5\n v_lshlrev_b32 %6, 16, %6\n v_lshlrev_b32 %7, 16, %7\n")
    if (MODE == 13) BODY("v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3\n v_lshl_add_u64 %4, %4, 1, %4\n v_lshl_add_u64 %5, %5, 1, %5\n v_lshl_add_u64 %6, %6, 1, %6\n v_lshl_add_u64 %7, %7, 1, %7\n")
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    // s_memtime counts at 100 MHz (constant clock) on this family: report raw ticks per instruction relative to v_add_f32
    printf("%-22s ticks/instr %.4f\n", name, s / 256 / (iters * 64.0 * 8));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("v_add_f32", out, cyc);
    run<1>("v_exp_f32", out, cyc);
    run<4>("v_rcp_f32", out, cyc);
    run<2>("v_exp_f16", out, cyc);
    run<3>("v_rcp_f16", out, cyc);
    run<5>("v_pk_add_f16", out, cyc);
    run<6>("v_pk_mul_f16", out, cyc);
    run<7>("v_fma_mixlo_f16", out, cyc);
    run<9>("v_fma_mix_f32", out, cyc);
    run<8>("v_cvt_pk_bf16_f32", out, cyc);
    run<10>("v_pk_add_f32", out, cyc);
    run<11>("v_pk_fma_f32", out, cyc);
    run<12>("v_cndmask_b32 vcc", out, cyc);
    run<16>("v_cndmask 0,v,vcc", out, cyc);
    run<14>("v_cndmask_e64 sgpr", out, cyc);
    run<15>("v_and_b32", out, cyc);
    run<17>("v_lshlrev_b32", out, cyc);
    return 0;
}
