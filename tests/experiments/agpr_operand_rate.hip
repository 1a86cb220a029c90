// Micro-benchmark (developer experiment): cost of feeding v_fmac_f64_dpp from AGPR-resident operands (v_accvgpr_read_b32 pairs, as the
// two-rows-per-lane mat-vec does for 31 of its 49 tiles) against VGPR-resident and LDS-resident operands. One wavefront, shader-clock cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
__global__ __launch_bounds__(64) void bench(double* out, long long* cyc, int iters) {
    __shared__ double lds[64 * 8];
    for (int i = 0; i < 8; ++i) lds[threadIdx.x * 8 + i] = 1.0 + i;
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, x = a0 * 0.5, w0 = 1.0000001, w1 = 1.0000002, w2 = 1.0000003;
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %1\n v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %1\n"
                 :: "v"(__double2loint(w0)), "v"(__double2hiint(w0)) : "a0", "a1", "a2", "a3", "a4", "a5");
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {   // VGPR operands
        asm volatile(REP16("v_fmac_f64_dpp %0, %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %3, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_fmac_f64_dpp %2, %3, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x), "v"(w0), "v"(w1), "v"(w2));
    }
    long long t1 = clock64();
    for (int it = 0; it < iters; ++it) {   // the loop's pattern: 6 AGPR reads, s_nop 0, 3 fmac
        asm volatile(REP16("v_accvgpr_read_b32 v40, a0\n v_accvgpr_read_b32 v41, a1\n v_accvgpr_read_b32 v42, a2\n v_accvgpr_read_b32 v43, a3\n v_accvgpr_read_b32 v44, a4\n v_accvgpr_read_b32 v45, a5\n s_nop 0\n"
                           "v_fmac_f64_dpp %0, %3, v[40:41] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %3, v[42:43] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_fmac_f64_dpp %2, %3, v[44:45] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x) : "v40", "v41", "v42", "v43", "v44", "v45");
    }
    long long t2 = clock64();
    for (int it = 0; it < iters; ++it) {   // AGPR reads alone
        asm volatile(REP16("v_accvgpr_read_b32 v40, a0\n v_accvgpr_read_b32 v41, a1\n v_accvgpr_read_b32 v42, a2\n v_accvgpr_read_b32 v43, a3\n v_accvgpr_read_b32 v44, a4\n v_accvgpr_read_b32 v45, a5\n")
                     ::: "v40", "v41", "v42", "v43", "v44", "v45");
    }
    long long t3 = clock64();
    const unsigned la = (unsigned)(size_t)(lds + threadIdx.x * 8) & 0xffff;
    for (int it = 0; it < iters; ++it) {   // LDS operands, one group ahead: ds_read_b128 x2 (4 doubles, 3 used), 3 fmac on the previous group's registers
        asm volatile(REP16("ds_read_b128 v[40:43], %4\n ds_read_b128 v[44:47], %4 offset:16\n"
                           "v_fmac_f64_dpp %0, %3, v[48:49] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %3, v[50:51] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_fmac_f64_dpp %2, %3, v[52:53] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "ds_read_b128 v[48:51], %4 offset:32\n ds_read_b128 v[52:55], %4 offset:48\n s_waitcnt lgkmcnt(2)\n"
                           "v_fmac_f64_dpp %0, %3, v[40:41] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %3, v[42:43] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_fmac_f64_dpp %2, %3, v[44:45] row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_waitcnt lgkmcnt(0)\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x), "v"(la) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "memory");
    }
    long long t4 = clock64();
    for (int it = 0; it < iters; ++it) {   // AGPR reads spread between the fmacs of the previous group (software pipelined)
        asm volatile(REP16("v_accvgpr_read_b32 v40, a0\n v_accvgpr_read_b32 v41, a1\n v_fmac_f64_dpp %0, %3, v[46:47] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_accvgpr_read_b32 v42, a2\n v_accvgpr_read_b32 v43, a3\n v_fmac_f64_dpp %1, %3, v[48:49] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_accvgpr_read_b32 v44, a4\n v_accvgpr_read_b32 v45, a5\n v_fmac_f64_dpp %2, %3, v[50:51] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_accvgpr_read_b32 v46, a0\n v_accvgpr_read_b32 v47, a1\n v_fmac_f64_dpp %0, %3, v[40:41] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_accvgpr_read_b32 v48, a2\n v_accvgpr_read_b32 v49, a3\n v_fmac_f64_dpp %1, %3, v[42:43] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                           "v_accvgpr_read_b32 v50, a4\n v_accvgpr_read_b32 v51, a5\n v_fmac_f64_dpp %2, %3, v[44:45] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
    }
    long long t5 = clock64();
    out[threadIdx.x] = a0 + a1 + a2;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
int main() {
    double* o; long long* c; (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&c, 64); (void)hipMemset(o, 0, 512);
    const int iters = 1000;
    hipLaunchKernelGGL(bench, dim3(1), dim3(64), 0, 0, o, c, iters);
    long long h[5]; (void)hipMemcpy(h, c, 40, hipMemcpyDeviceToHost);
    const char* nm[5] = {"3 fmac_dpp, VGPR operands", "6 accvgpr_read + s_nop + 3 fmac_dpp", "6 accvgpr_read", "2 ds_read_b128 + 3 fmac_dpp (prefetched)", "6 accvgpr_read interleaved with 3 fmac_dpp (pipelined)"};
    const double div[5] = {16, 16, 16, 32, 32};
    for (int i = 0; i < 5; ++i) printf("%-58s %.1f cycles per group of 3 fmac\n", nm[i], (double)h[i] / (iters * div[i]));
    return 0;
}
