// polympc_amd — batched boxADMM::solve on the two-rows-per-lane register path (65..112 KKT rows, pmpc_qp_reg2.hpp): the QP entry point's
// specialisations for the sizes of config B (CSTR, 11 nodes: 66 + 44), of the reference's 11-node robot grid (55 + 33) and of the other robot / CSTR grids the
// fused SQP kernel serves on this path (9, 10, 12, 13 nodes).
#include <hip/hip_runtime.h>
#include "../../include/polympc_amd.h"
#include "pmpc_qp_reg2.hpp"

using namespace pmpc;

template <int NN, int MM>
__global__ __launch_bounds__(64, 1) void qp_boxadmm_reg2_kernel(int B, const double* __restrict__ H, const double* __restrict__ h,
                                                                const double* __restrict__ A, const double* __restrict__ Alb,
                                                                const double* __restrict__ Aub, const double* __restrict__ xlb,
                                                                const double* __restrict__ xub, const double* __restrict__ x0,
                                                                const double* __restrict__ y0, pmpc_qp_settings s,
                                                                double* __restrict__ x, double* __restrict__ y, pmpc_qp_info* __restrict__ info) {
    __shared__ __attribute__((aligned(16))) double tr[RegKkt2<NN + MM>::TRI];
    const int b = blockIdx.x;
    if (b >= B) return;
    pmpc_qp_info qi;
    boxadmm_solve_reg2<NN, MM, false, true>(H + (size_t)b * NN * NN, h + (size_t)b * NN, A + (size_t)b * MM * NN, Alb + (size_t)b * MM, Aub + (size_t)b * MM,
                               xlb + (size_t)b * NN, xub + (size_t)b * NN, x0 ? x0 + (size_t)b * NN : nullptr,
                               y0 ? y0 + (size_t)b * (NN + MM) : nullptr, s, qi, x + (size_t)b * NN, y + (size_t)b * (NN + MM), tr);
    if (lane_id() == 0) info[b] = qi;
}

// returns 1 when (n, m) has a specialisation and the kernel was launched on `stream`, 0 otherwise, -1 on a launch error
extern "C" int pmpc_internal_qp_reg2_launch(void* stream, int B, int n, int m, const double* H, const double* h, const double* A, const double* Alb,
                                            const double* Aub, const double* xlb, const double* xub, const double* x0, const double* y0,
                                            const pmpc_qp_settings* s, double* x, double* y, pmpc_qp_info* info) {
#define PMPC_REG2_CASE(NN_, MM_)                                                                                                              \
    if (n == NN_ && m == MM_) {                                                                                                                \
        hipLaunchKernelGGL((qp_boxadmm_reg2_kernel<NN_, MM_>), dim3(B), dim3(WAVE), 0, (hipStream_t)stream, B, H, h, A, Alb, Aub, xlb, xub, x0, \
                           y0, *s, x, y, info);                                                                                                \
        return hipGetLastError() == hipSuccess ? 1 : -1;                                                                                       \
    }
    PMPC_REG2_CASE(66, 44)
    PMPC_REG2_CASE(55, 33)
    PMPC_REG2_CASE(45, 27)   // robot grids of 9, 10, 12 and 13 nodes
    PMPC_REG2_CASE(50, 30)
    PMPC_REG2_CASE(60, 36)
    PMPC_REG2_CASE(65, 39)
    PMPC_REG2_CASE(54, 36)   // CSTR grids of 9 and 10 nodes
    PMPC_REG2_CASE(60, 40)
    PMPC_REG2_CASE(80, 48)   // 113..128 rows (8 x 8 tiles, last tile row / column in LDS): robot grids of 15 and 16 nodes (the reference's mpc_wrapper_test grid), CSTR of 12
    PMPC_REG2_CASE(75, 45)
    PMPC_REG2_CASE(72, 48)
#undef PMPC_REG2_CASE
    return 0;
}
