// tools/ratebench.hip — issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the rollout
// kernels are made of, measured on the device: N independent chains of one instruction per wave, W waves per SIMD,
// wall time -> cycles at the measured clock.  Grounds the cost model in DESIGN.md §4 (is Philox's v_mad_u64_u32 a
// quarter-rate instruction? what do the fp64 division helpers cost?).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define COMMA ,
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// Each kernel runs `iters` iterations of 64 instances of the instruction over 4 independent register chains.
#define KERNEL(name, decl, body, sink)                                   \
    __global__ void __launch_bounds__(256) name(int iters, double *out) { \
        decl;                                                            \
        for (int i = 0; i < iters; ++i) {                                \
            REP16(body)                                                  \
        }                                                                \
        sink;                                                            \
    }

#define D4 double a = threadIdx.x * 1e-3 + 1.0, b = a + 1.0, c = a + 2.0, d = a + 3.0, k = 1.0000001
#define DSINK if (a + b + c + d == 12345.678) out[0] = a
#define U4 uint32_t a = threadIdx.x + 1, b = a + 1, c = a + 2, d = a + 3; uint64_t pa = a, pb = b, pc = c, pd = d
#define USINK if (a + b + c + d + (uint32_t)pa + (uint32_t)pb + (uint32_t)pc + (uint32_t)pd == 12345u) out[0] = a

KERNEL(k_fma_f64, D4,
       asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));, DSINK)
KERNEL(k_mul_f64, D4,
       asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));, DSINK)
KERNEL(k_add_f64, D4,
       asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));, DSINK)
KERNEL(k_rcp_f64, D4,
       asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));, DSINK)
KERNEL(k_div_fixup_f64, D4,
       asm volatile("v_div_fixup_f64 %0, %0, %4, %4\n v_div_fixup_f64 %1, %1, %4, %4\n v_div_fixup_f64 %2, %2, %4, %4\n v_div_fixup_f64 %3, %3, %4, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));, DSINK)
KERNEL(k_div_scale_f64, D4,
       asm volatile("v_div_scale_f64 %0, vcc, %0, %4, %4\n v_div_scale_f64 %1, vcc, %1, %4, %4\n v_div_scale_f64 %2, vcc, %2, %4, %4\n v_div_scale_f64 %3, vcc, %3, %4, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k) : "vcc");, DSINK)
KERNEL(k_cvt_f32_f64, D4; float fa = 0 COMMA fb = 0 COMMA fc = 0 COMMA fd = 0,
       asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7"
                    : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(a), "v"(b), "v"(c), "v"(d));, if (fa + fb + fc + fd == 1234.5f) out[0] = a)
KERNEL(k_mad_u64_u32, U4,
       asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0"
                    : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(0xD2511F53u) : "vcc");
       a ^= (uint32_t)pa; b ^= (uint32_t)pb; c ^= (uint32_t)pc; d ^= (uint32_t)pd;, USINK)
KERNEL(k_xor_only, U4,
       asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0xD2511F53u));, USINK)
KERNEL(k_mul_lo_u32, U4,
       asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0xD2511F53u));, USINK)
KERNEL(k_mul_hi_u32, U4,
       asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0xD2511F53u));, USINK)
KERNEL(k_mul_u32_u24, U4,
       asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0x511F53u));, USINK)
KERNEL(k_fma_f32, U4; float fa = 1.f COMMA fb = 2.f COMMA fc = 3.f COMMA fd = 4.f,
       asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                    : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(1.0000001f));, if (fa + fb + fc + fd == 1234.5f) out[0] = a)

template <typename K>
void run(const char *name, K kern, int per_iter_extra, double *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, out);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double instr_per_simd = (double)iters * 64 * waves_per_simd;  // wave-instructions issued on one SIMD
        printf("%-18s waves/SIMD=%d  %8.3f ms  %6.2f ns per wave-instr per SIMD (x2.4 GHz = %5.2f cycles)\n", name,
               waves_per_simd, best, best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.4);
    }
}

int main() {
    double *out;
    hipMalloc(&out, 64);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device clock attribute: %d kHz\n", clk);
    // spin-up
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_fma_f64, dim3(2048), dim3(256), 0, 0, 2000, out);
    hipDeviceSynchronize();
    run("v_fma_f64", k_fma_f64, 0, out);
    run("v_mul_f64", k_mul_f64, 0, out);
    run("v_add_f64", k_add_f64, 0, out);
    run("v_rcp_f64", k_rcp_f64, 0, out);
    run("v_div_scale_f64", k_div_scale_f64, 0, out);
    run("v_div_fixup_f64", k_div_fixup_f64, 0, out);
    run("v_cvt_f32_f64", k_cvt_f32_f64, 0, out);
    run("v_fma_f32", k_fma_f32, 0, out);
    run("v_xor_b32", k_xor_only, 0, out);
    run("v_mad_u64_u32+xor", k_mad_u64_u32, 0, out);
    run("v_mul_lo_u32", k_mul_lo_u32, 0, out);
    run("v_mul_hi_u32", k_mul_hi_u32, 0, out);
    run("v_mul_u32_u24", k_mul_u32_u24, 0, out);
    return 0;
}
