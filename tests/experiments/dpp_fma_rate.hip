// Micro-benchmark (developer experiment): issue cost of v_fmac_f64 with a DPP row_newbcast operand against the plain
// v_fma_f64, v_readlane-fed fma, and the permlane swaps, one wavefront, shader-clock cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
__global__ void bench(double* out, long long* cyc, int iters) {
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, x = a0 * 0.5, w = 1.0000001;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                           "v_fmac_f64_dpp %2, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(w));
    }
    long long t1 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP16("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(w));
    }
    long long t2 = clock64();
    for (int it = 0; it < iters; ++it) {   // one chain only: dependent-issue latency
        asm volatile(REP16("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n v_fmac_f64 %0, %4, %5\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(w));
    }
    long long t3 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                           "v_fmac_f64_dpp %0, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(w));
    }
    long long t4 = clock64();
    for (int it = 0; it < iters; ++it) {   // readlane-fed: 2 v_readlane + fma with SGPR operand, 4 chains
        asm volatile(REP16("v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n s_nop 0\n v_fma_f64 %0, s[20:21], %6, %0\n"
                           "v_readlane_b32 s22, %4, 5\n v_readlane_b32 s23, %5, 5\n s_nop 0\n v_fma_f64 %1, s[22:23], %6, %1\n"
                           "v_readlane_b32 s24, %4, 7\n v_readlane_b32 s25, %5, 7\n s_nop 0\n v_fma_f64 %2, s[24:25], %6, %2\n"
                           "v_readlane_b32 s26, %4, 9\n v_readlane_b32 s27, %5, 9\n s_nop 0\n v_fma_f64 %3, s[26:27], %6, %3\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(__double2loint(x)), "v"(__double2hiint(x)), "v"(w) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    }
    long long t5 = clock64();
    int p0 = __double2loint(a0), p1 = __double2loint(a1);
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP16("v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %0, %1\n") : "+v"(p0), "+v"(p1));
    }
    long long t6 = clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + p0 + p1;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; }
}
int main() {
    double* o; long long* c; (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&c, 64); (void)hipMemset(o, 0, 512);
    const int iters = 1000;
    hipLaunchKernelGGL(bench, dim3(1), dim3(64), 0, 0, o, c, iters);
    long long h[6]; (void)hipMemcpy(h, c, 48, hipMemcpyDeviceToHost);
    const char* nm[6] = {"v_fmac_f64_dpp x4 chains", "v_fmac_f64 x4 chains", "v_fmac_f64 1 chain", "v_fmac_f64_dpp 1 chain", "2 readlane + fma (x4 chains), per group", "permlane swap b32"};
    for (int i = 0; i < 6; ++i) printf("%-42s %.2f cycles per instruction%s\n", nm[i], (double)h[i] / (iters * 64.0), i == 4 ? " group" : "");
    return 0;
}
