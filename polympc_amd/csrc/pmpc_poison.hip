// polympc_amd — device poisoning (developer harness, PMPC_POISON=1 / pmpc_debug_set_poison): before a launch of any product kernel every place an
// UNINITIALISED read could fetch stale-but-plausible data from is filled with signalling NaNs, without touching the product kernels themselves —
// the binaries under test are the shipped ones:
//   * every CU's LDS (one workgroup per CU with the opt-in maximum of dynamic LDS — LDS is not cleared between workgroups),
//   * every SIMD's whole register file, architected and accumulation (one 512-register wavefront per SIMD — registers are not cleared between
//     wavefronts; a wavefront of the next kernel that reads a register it never wrote sees the pattern, as do lanes that were inactive when a
//     register was written),
//   * the low 4 KB per lane of the private segment (scratch) of the wave slots the poison kernel occupied (a reload of a spill slot that was
//     stored under a partial EXEC mask — DESIGN.md, compiler hazard 3 — then returns the pattern in the lanes that were inactive),
//   * the HBM workspace and the staging buffers of the host-buffer entry points (pmpc_context.hpp: ensure_ws / ensure_scratch).
// Pattern: the dword 0x7FF47FF4 everywhere — any aligned or misaligned pair of dwords is the fp64 signalling NaN 0x7FF47FF47FF47FF4, a single dword an
// fp32 NaN. A kernel that reads what it never wrote returns NaN (PMPC_FLAG_NONFINITE / a failed bit-exactness test) instead of a run-to-run
// different number. tests: `PMPC_POISON=1 python -m pytest tests -m gpu`, tests/tools_soak_routes.py.
#include <hip/hip_runtime.h>
#include "pmpc_context.hpp"

#define PMPC_X1(p, n) p #n ", %0\n\t"
#define PMPC_X10(p, t) PMPC_X1(p, t##0) PMPC_X1(p, t##1) PMPC_X1(p, t##2) PMPC_X1(p, t##3) PMPC_X1(p, t##4) PMPC_X1(p, t##5) PMPC_X1(p, t##6) PMPC_X1(p, t##7) PMPC_X1(p, t##8) PMPC_X1(p, t##9)
#define PMPC_X256(p) PMPC_X1(p, 0) PMPC_X1(p, 1) PMPC_X1(p, 2) PMPC_X1(p, 3) PMPC_X1(p, 4) PMPC_X1(p, 5) PMPC_X1(p, 6) PMPC_X1(p, 7) PMPC_X1(p, 8) PMPC_X1(p, 9) \
    PMPC_X10(p, 1) PMPC_X10(p, 2) PMPC_X10(p, 3) PMPC_X10(p, 4) PMPC_X10(p, 5) PMPC_X10(p, 6) PMPC_X10(p, 7) PMPC_X10(p, 8) PMPC_X10(p, 9) \
    PMPC_X10(p, 10) PMPC_X10(p, 11) PMPC_X10(p, 12) PMPC_X10(p, 13) PMPC_X10(p, 14) PMPC_X10(p, 15) PMPC_X10(p, 16) PMPC_X10(p, 17) PMPC_X10(p, 18) PMPC_X10(p, 19) \
    PMPC_X10(p, 20) PMPC_X10(p, 21) PMPC_X10(p, 22) PMPC_X10(p, 23) PMPC_X10(p, 24) PMPC_X1(p, 250) PMPC_X1(p, 251) PMPC_X1(p, 252) PMPC_X1(p, 253) PMPC_X1(p, 254) PMPC_X1(p, 255)
#define PMPC_C1(p, n) p #n,
#define PMPC_C10(p, t) PMPC_C1(p, t##0) PMPC_C1(p, t##1) PMPC_C1(p, t##2) PMPC_C1(p, t##3) PMPC_C1(p, t##4) PMPC_C1(p, t##5) PMPC_C1(p, t##6) PMPC_C1(p, t##7) PMPC_C1(p, t##8) PMPC_C1(p, t##9)
#define PMPC_C256(p) PMPC_C1(p, 0) PMPC_C1(p, 1) PMPC_C1(p, 2) PMPC_C1(p, 3) PMPC_C1(p, 4) PMPC_C1(p, 5) PMPC_C1(p, 6) PMPC_C1(p, 7) PMPC_C1(p, 8) PMPC_C1(p, 9) \
    PMPC_C10(p, 1) PMPC_C10(p, 2) PMPC_C10(p, 3) PMPC_C10(p, 4) PMPC_C10(p, 5) PMPC_C10(p, 6) PMPC_C10(p, 7) PMPC_C10(p, 8) PMPC_C10(p, 9) \
    PMPC_C10(p, 10) PMPC_C10(p, 11) PMPC_C10(p, 12) PMPC_C10(p, 13) PMPC_C10(p, 14) PMPC_C10(p, 15) PMPC_C10(p, 16) PMPC_C10(p, 17) PMPC_C10(p, 18) PMPC_C10(p, 19) \
    PMPC_C10(p, 20) PMPC_C10(p, 21) PMPC_C10(p, 22) PMPC_C10(p, 23) PMPC_C10(p, 24) PMPC_C1(p, 250) PMPC_C1(p, 251) PMPC_C1(p, 252) PMPC_C1(p, 253) PMPC_C1(p, 254) PMPC_C1(p, 255)

constexpr unsigned POISON_DWORD = 0x7FF47FF4u;
constexpr int POISON_SCRATCH_DWORDS = 1024;   // 4 KB per lane

// one workgroup of four wavefronts per CU (the whole LDS of a CU as its dynamic allocation, 512 registers per wavefront: nothing else fits beside it)
__global__ __launch_bounds__(256, 1) void poison_device_kernel(unsigned lds_dwords, long long spin_cycles, unsigned* __restrict__ sink) {
    extern __shared__ unsigned lds_all[];
    for (unsigned i = threadIdx.x; i < lds_dwords; i += 256) lds_all[i] = POISON_DWORD;
    unsigned priv[POISON_SCRATCH_DWORDS];
    for (int i = 0; i < POISON_SCRATCH_DWORDS; ++i) { unsigned v = POISON_DWORD; asm volatile("" : "+v"(v)); priv[i] = v; }
    unsigned chk = 0;
    for (int i = threadIdx.x & 7; i < POISON_SCRATCH_DWORDS; i += 8) chk |= priv[i] ^ POISON_DWORD;   // (keeps the private array in scratch memory)
    if (chk != 0 && sink) sink[0] = chk;
    // stay resident until every CU holds its workgroup (otherwise a CU that finished early would take a second one and another CU none)
    const long long t0 = clock64();
    while (clock64() - t0 < spin_cycles) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    unsigned pat = POISON_DWORD;
    asm volatile(PMPC_X256("v_mov_b32 v") PMPC_X256("v_accvgpr_write_b32 a") "s_nop 0"
                 :: "s"(pat)
                 : PMPC_C256("v") PMPC_C256("a") "memory");
}

extern "C" pmpc_status pmpc_internal_poison_device(pmpc_context* ctx) {
    if (!ctx) return PMPC_ERR_INVALID_ARGUMENT;
    if (!ctx->poison) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t lds = ctx->lds_limit_device;
    HIPCHK(hipFuncSetAttribute((const void*)poison_device_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int cus = ctx->simd_count / 4;
    hipLaunchKernelGGL(poison_device_kernel, dim3(cus), dim3(256), lds, ctx->stream, (unsigned)(lds / 4), (long long)200000, (unsigned*)nullptr);
    HIPCHK(hipGetLastError());
    return PMPC_OK;
}
