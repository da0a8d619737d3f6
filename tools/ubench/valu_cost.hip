// valu_cost.hip — issue cost (cycles per wave-instruction per SIMD) of the FP64 VALU ops the residual
// sweep uses, measured with 8 waves/SIMD of independent chains (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP> __global__ void __launch_bounds__(256) k(double* out, double a, double b, int iters)
{
    double x0 = a + threadIdx.x, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    unsigned int u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3;
    float f0 = (float)a + threadIdx.x, f1 = (float)b, f2 = (float)a, f3 = 1.5f, f4 = 2.5f, f5 = 3.5f;
    for (int i = 0; i < iters; ++i) {
#define REP8(S) S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
        if (OP == 0) {
#define S0(v) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(a));
            REP8(S0)
        } else if (OP == 1) {
#define S1(v) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v) : "v"(b));
            REP8(S1)
        } else if (OP == 2) {
#define S2(v) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(b));
            REP8(S2)
        } else if (OP == 3) {
#define S3(v) asm volatile("v_rcp_f64 %0, %0" : "+v"(v));
            REP8(S3)
        } else if (OP == 4) {
#define S4(v) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(v), "v"(b) : "vcc");
            REP8(S4)
        } else if (OP == 5) {
#define S5(v) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(v) : "v"(b) : "vcc");
            REP8(S5)
        } else if (OP == 6) {
#define S6(v) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(a));
            REP8(S6)
        } else if (OP == 7) {
#define S7(v) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(a) : "vcc");
            REP8(S7)
        } else if (OP == 8) {
#define S8(v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(u3));
            S8(u0) S8(u1) S8(u2) S8(u0) S8(u1) S8(u2) S8(u0) S8(u1)
        } else if (OP == 9) {
#define S9(v) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v) : "v"(u3));
            S9(u0) S9(u1) S9(u2) S9(u0) S9(u1) S9(u2) S9(u0) S9(u1)
        } else if (OP == 10) {
#define S10(v) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(v), "v"(u3) : "vcc");
            S10(u0) S10(u1) S10(u2) S10(u0) S10(u1) S10(u2) S10(u0) S10(u1)
        } else if (OP == 11) {
#define S11(v) asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(u3));
            S11(u0) S11(u1) S11(u2) S11(u0) S11(u1) S11(u2) S11(u0) S11(u1)
        } else if (OP == 12) {
#define S12(v) asm volatile("v_sqrt_f64 %0, %0" : "+v"(v));
            REP8(S12)
        } else if (OP == 13) {
#define S13(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(f1), "v"(f2));
            S13(f0) S13(f3) S13(f4) S13(f5) S13(f0) S13(f3) S13(f4) S13(f5)
        } else if (OP == 14) {
#define S14(v) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v) : "v"(f1), "v"(f2));
            S14(f0) S14(f3) S14(f4) S14(f5) S14(f0) S14(f3) S14(f4) S14(f5)
        } else if (OP == 15) {
#define S15(v) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(b), "v"(a));
            REP8(S15)
        } else if (OP == 16) {
#define S16(v) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
            S16(f0) S16(f3) S16(f4) S16(f5) S16(f0) S16(f3) S16(f4) S16(f5)
        } else if (OP == 17) {
#define S17(v) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(v) : "v"(f1));
            S17(f0) S17(f3) S17(f4) S17(f5) S17(f0) S17(f3) S17(f4) S17(f5)
        } else if (OP == 18) {
#define S18(v) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(v), "v"(f1) : "vcc");
            S18(f0) S18(f3) S18(f4) S18(f5) S18(f0) S18(f3) S18(f4) S18(f5)
        } else if (OP == 19) {
#define S19(v) asm volatile("v_fma_f32 %0, %0, s4, %1" : "+v"(v) : "v"(f2) : "s4");
            S19(f0) S19(f3) S19(f4) S19(f5) S19(f0) S19(f3) S19(f4) S19(f5)
        } else if (OP == 20) {
#define S20(v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(f1));
            S20(f0) S20(f3) S20(f4) S20(f5) S20(f0) S20(f3) S20(f4) S20(f5)
        } else if (OP == 21) {
#define S21(v) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(b));
            REP8(S21)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + u0 + u1 + u2 + f0 + f3 + f4 + f5;
}

template <int OP> int run(const char* name, double* out)
{
    const int iters = 20000, blocks = 256 * 8;        // 8 blocks of 4 waves per CU = 8 waves/SIMD
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.000001, 0.999999, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.000001, 0.999999, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // wave-instructions per SIMD = 8 waves * iters * 8 ; cycles = ms * f
    const double winst = 8.0 * iters * 8.0;
    printf("%-18s %8.3f ms  -> %6.2f ns per wave-instr per SIMD  (= %5.2f cycles @2.4GHz, %5.2f @2.0GHz)\n", name, ms,
           ms * 1e6 / winst, ms * 1e6 / winst * 2.4, ms * 1e6 / winst * 2.0);
    return 0;
}

int main()
{
    double* out;
    CK(hipMalloc(&out, 256 * 8 * 256 * 8));
    run<0>("v_fma_f64", out); run<1>("v_mul_f64", out); run<2>("v_add_f64", out); run<3>("v_rcp_f64", out);
    run<4>("v_cmp_lt_f64", out); run<5>("v_div_scale_f64", out); run<6>("v_div_fixup_f64", out);
    run<7>("v_div_fmas_f64", out); run<8>("v_add_u32", out); run<9>("v_alignbit_b32", out);
    run<10>("v_cmp_lt_u32", out); run<11>("v_mov_b32", out); run<12>("v_sqrt_f64", out);
    run<13>("v_fma_f32", out); run<14>("v_fmac_f32", out); run<15>("v_pk_fma_f32", out); run<16>("v_rcp_f32", out);
    run<17>("v_max_f32 |a|,|b|", out); run<18>("v_cmp_gt_f32", out); run<19>("v_fma_f32 (sgpr)", out); run<20>("v_mul_f32", out);
    run<21>("v_pk_mul_f32", out);
    return 0;
}
