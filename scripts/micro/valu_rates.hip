// Microbenchmark: issue cost of the vector instructions the raster path is made of.
// Every lane runs a long loop over eight independent accumulators of one instruction kind (inline assembly, so the
// compiler neither fuses nor vectorises anything); the grid fills every SIMD with four wavefronts.  Reported: wave
// instructions per SIMD per microsecond and the cycles one instruction occupies a SIMD at 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/micro/valu_rates scripts/micro/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 4096, kAcc = 8;

#define BODY32(OPSTR)                                                                        \
    float a[kAcc];                                                                           \
    for (int k = 0; k < kAcc; k++) a[k] = float(threadIdx.x + k);                            \
    const float b = 1.0000001f, c = 0.5f;                                                    \
    for (int it = 0; it < kIters; it++)                                                      \
        _Pragma("unroll") for (int k = 0; k < kAcc; k++) asm volatile(OPSTR : "+v"(a[k]) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
    float s = 0;                                                                             \
    for (int k = 0; k < kAcc; k++) s += a[k];                                                \
    if (s == 12345.678f) out[0] = s;

__global__ void k_mul_f32(float *out) { BODY32("v_mul_f32 %0, %0, %1") }
__global__ void k_fma_f32(float *out) { BODY32("v_fma_f32 %0, %0, %1, %2") }
__global__ void k_rcp_f32(float *out) { BODY32("v_rcp_f32 %0, %0") }
__global__ void k_sqrt_f32(float *out) { BODY32("v_sqrt_f32 %0, %0") }
__global__ void k_mov_b32(float *out) { BODY32("v_mov_b32 %0, %1") }
__global__ void k_cndmask(float *out) { BODY32("v_cndmask_b32 %0, %0, %1, vcc") }
__global__ void k_cmp_f32(float *out) { BODY32("v_cmp_lt_f32 vcc, %0, %1") }
__global__ void k_cmp_u32(float *out) { BODY32("v_cmp_eq_u32 vcc, %0, %1") }
__global__ void k_cmp_u32_sgpr(float *out) { BODY32("v_cmp_eq_u32_e64 s[20:21], %0, %1") }
__global__ void k_cndmask_sgpr(float *out) { BODY32("v_cndmask_b32_e64 %0, %0, %1, s[20:21]") }
__global__ void k_addc(float *out) { BODY32("v_addc_co_u32 %0, vcc, %0, %1, vcc") }
__global__ void k_cmp_then_cnd(float *out) { BODY32("v_cmp_eq_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc") }
__global__ void k_cmp_then_addc(float *out) { BODY32("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc") }
__global__ void k_add_u32(float *out) { BODY32("v_add_u32 %0, %0, %1") }
__global__ void k_and_b32(float *out) { BODY32("v_and_b32 %0, %0, %1") }
__global__ void k_bfe_u32(float *out) { BODY32("v_bfe_u32 %0, %0, 8, 8") }
__global__ void k_add3_u32(float *out) { BODY32("v_add3_u32 %0, %0, %1, %2") }
__global__ void k_max3_f32(float *out) { BODY32("v_max3_f32 %0, %0, %1, %2") }
__global__ void k_mul_lo_u32(float *out) { BODY32("v_mul_lo_u32 %0, %0, %1") }
__global__ void k_cvt_i32_f32(float *out) { BODY32("v_cvt_i32_f32 %0, %0") }
__global__ void k_cvt_f32_u32(float *out) { BODY32("v_cvt_f32_u32 %0, %0") }
__global__ void k_min_u32(float *out) { BODY32("v_min_u32 %0, %0, %1") }
__global__ void k_max_f32(float *out) { BODY32("v_max_f32 %0, %0, %1") }
__global__ void k_xor_b32(float *out) { BODY32("v_xor_b32 %0, %0, %1") }
__global__ void k_sub_u32(float *out) { BODY32("v_sub_u32 %0, %0, %1") }
__global__ void k_lshrrev_b32(float *out) { BODY32("v_lshrrev_b32 %0, 8, %0") }
__global__ void k_fmac_f32(float *out) { BODY32("v_fmac_f32 %0, %1, %2") }
__global__ void k_add_f32(float *out) { BODY32("v_add_f32 %0, %0, %1") }
__global__ void k_floor_f32(float *out) { BODY32("v_floor_f32 %0, %0") }
__global__ void k_bcnt(float *out) { BODY32("v_bcnt_u32_b32 %0, %1, %0") }
__global__ void k_alignbyte(float *out) { BODY32("v_alignbyte_b32 %0, %0, %1, 1") }
__global__ void k_perm(float *out) { BODY32("v_perm_b32 %0, %0, %1, %2") }
__global__ void k_and_or(float *out) { BODY32("v_and_or_b32 %0, %0, %1, %2") }
__global__ void k_lshl_add_u32(float *out) { BODY32("v_lshl_add_u32 %0, %0, 2, %1") }
__global__ void k_cmp_sdwa(float *out) { BODY32("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:BYTE_2 src1_sel:DWORD") }
__global__ void k_and_sdwa(float *out) { BODY32("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") }
__global__ void k_div_scale(float *out) { BODY32("v_div_scale_f32 %0, vcc, %0, %1, %0") }
__global__ void k_div_fmas(float *out) { BODY32("v_div_fmas_f32 %0, %0, %1, %2") }
__global__ void k_div_fixup(float *out) { BODY32("v_div_fixup_f32 %0, %0, %1, %2") }
__global__ void k_mul_f32_sgpr(float *out) { BODY32("v_mul_f32 %0, s20, %0") }
__global__ void k_mul_f32_lit(float *out) { BODY32("v_mul_f32 %0, 0x3f8ccccd, %0") }
__global__ void k_mul_f32_inl(float *out) { BODY32("v_mul_f32 %0, 0.5, %0") }
__global__ void k_add_f32_inl(float *out) { BODY32("v_add_f32 %0, -0.5, %0") }
__global__ void k_fma_f32_sgpr(float *out) { BODY32("v_fma_f32 %0, %0, s20, %1") }
__global__ void k_fma_f32_abs(float *out) { BODY32("v_fma_f32 %0, |%0|, %1, %2") }
__global__ void k_add_f32_neg(float *out) { BODY32("v_sub_f32 %0, %0, %1") }
__global__ void k_add_f32_e64abs(float *out) { BODY32("v_add_f32_e64 %0, |%0|, %1") }
__global__ void k_cndmask_vcc_after(float *out) { BODY32("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc") }
__global__ void k_mbcnt(float *out) { BODY32("v_mbcnt_lo_u32_b32 %0, -1, %0") }
__global__ void k_readlane(float *out) { BODY32("v_readlane_b32 s20, %0, 3") }

#define BODY64(OPSTR)                                                                        \
    double a[kAcc];                                                                          \
    for (int k = 0; k < kAcc; k++) a[k] = double(threadIdx.x + k);                           \
    const double b = 1.0000001, c = 0.5;                                                     \
    for (int it = 0; it < kIters; it++)                                                      \
        _Pragma("unroll") for (int k = 0; k < kAcc; k++) asm volatile(OPSTR : "+v"(a[k]) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
    double s = 0;                                                                            \
    for (int k = 0; k < kAcc; k++) s += a[k];                                                \
    if (s == 12345.678) out[0] = float(s);

__global__ void k_mul_f64(float *out) { BODY64("v_mul_f64 %0, %0, %1") }
__global__ void k_add_f64(float *out) { BODY64("v_add_f64 %0, %0, %1") }
__global__ void k_fma_f64(float *out) { BODY64("v_fma_f64 %0, %0, %1, %2") }
__global__ void k_pk_mul_f32(float *out) { BODY64("v_pk_mul_f32 %0, %0, %1") }
__global__ void k_pk_fma_f32(float *out) { BODY64("v_pk_fma_f32 %0, %0, %1, %2") }
__global__ void k_pk_add_f32(float *out) { BODY64("v_pk_add_f32 %0, %0, %1") }
__global__ void k_cmp_f64(float *out) { BODY64("v_cmp_lt_f64 vcc, %0, %1") }
__global__ void k_lshl_b64(float *out) { BODY64("v_lshlrev_b64 %0, 1, %0") }

__global__ void k_cvt_f64_f32(float *out) {
    float a[kAcc];
    double d[kAcc];
    for (int k = 0; k < kAcc; k++) a[k] = float(threadIdx.x + k);
    for (int it = 0; it < kIters; it++)
#pragma unroll
        for (int k = 0; k < kAcc; k++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
    double s = 0;
    for (int k = 0; k < kAcc; k++) s += d[k];
    if (s == 12345.678) out[0] = float(s);
}

__global__ void k_cvt_f32_f64(float *out) {
    float a[kAcc];
    double d[kAcc];
    for (int k = 0; k < kAcc; k++) d[k] = double(threadIdx.x + k);
    for (int it = 0; it < kIters; it++)
#pragma unroll
        for (int k = 0; k < kAcc; k++) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]));
    float s = 0;
    for (int k = 0; k < kAcc; k++) s += a[k];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int simds = 256 * 4, waves_per_simd = 4;
    const int wgs = simds * waves_per_simd / 4;  // 256 threads = 4 wavefronts per workgroup
    struct { const char *name; void (*fn)(float *); } tests[] = {
        {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_mov_b32", k_mov_b32}, {"v_cndmask_b32", k_cndmask},
        {"v_cmp_lt_f32", k_cmp_f32}, {"v_cmp_eq_u32", k_cmp_u32}, {"v_cmp_eq_u32 sgpr", k_cmp_u32_sgpr},
        {"v_cndmask sgpr", k_cndmask_sgpr}, {"v_addc_co_u32", k_addc}, {"cmp+cndmask", k_cmp_then_cnd}, {"cmp+addc", k_cmp_then_addc},
        {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32}, {"v_bfe_u32", k_bfe_u32}, {"v_add3_u32", k_add3_u32},
        {"v_max3_f32", k_max3_f32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_min_u32", k_min_u32}, {"v_max_f32", k_max_f32}, {"v_xor_b32", k_xor_b32}, {"v_sub_u32", k_sub_u32}, {"v_lshrrev_b32", k_lshrrev_b32},
        {"v_fmac_f32", k_fmac_f32}, {"v_add_f32", k_add_f32}, {"v_floor_f32", k_floor_f32}, {"v_bcnt_u32_b32", k_bcnt},
        {"v_alignbyte_b32", k_alignbyte}, {"v_perm_b32", k_perm}, {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add_u32},
        {"v_cmp_eq_u32 sdwa", k_cmp_sdwa}, {"v_and_b32 sdwa", k_and_sdwa}, {"v_div_scale_f32", k_div_scale}, {"v_div_fmas_f32", k_div_fmas},
        {"v_div_fixup_f32", k_div_fixup}, {"v_mul_f32 sgpr", k_mul_f32_sgpr}, {"v_mul_f32 literal", k_mul_f32_lit}, {"v_mul_f32 inline", k_mul_f32_inl},
        {"v_add_f32 inline", k_add_f32_inl}, {"v_fma_f32 sgpr", k_fma_f32_sgpr}, {"v_fma_f32 |abs|", k_fma_f32_abs}, {"v_sub_f32", k_add_f32_neg},
        {"v_add_f32 e64 |abs|", k_add_f32_e64abs}, {"2x v_cndmask vcc", k_cndmask_vcc_after}, {"v_mbcnt_lo", k_mbcnt}, {"v_readlane_b32", k_readlane},
        {"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32},
        {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_fma_f64", k_fma_f64}, {"v_cmp_lt_f64", k_cmp_f64},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_lshlrev_b64", k_lshl_b64},
    };
    printf("%-18s %10s %14s %12s\n", "instruction", "us", "inst/SIMD/us", "cycles@2.4GHz");
    for (auto &t : tests) {
        hipLaunchKernelGGL(t.fn, dim3(wgs), dim3(256), 0, 0, out);  // warm-up
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(t.fn, dim3(wgs), dim3(256), 0, 0, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double insts_per_simd = double(kIters) * kAcc * waves_per_simd;
        const double us = ms * 1e3;
        printf("%-18s %10.1f %14.1f %12.2f\n", t.name, us, insts_per_simd / us, us * 2400.0 / insts_per_simd);
    }
    return 0;
}
