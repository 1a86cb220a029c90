// polympc_amd — batched boxADMM::solve with the KKT factor in HBM (pmpc_qp_big.hpp: blocked left-looking tile LDL^T with fp64 MFMA updates,
// 16-column blocked substitutions): the QP entry point for systems of BIG_KKT_MIN_ROWS rows and more, and for everything whose packed triangle does not fit
// LDS (box_admm.hpp:88-205 at the reference's kite size, n + m = 464). One wavefront per QP, 13 KB of LDS (x, y, the right-hand side and the tile
// pipeline's slots); the other ADMM vectors live behind the factor in the per-QP HBM workspace.
#include <hip/hip_runtime.h>
#include "../../include/polympc_amd.h"
#include "pmpc_qp_big.hpp"

using namespace pmpc;

__global__ __launch_bounds__(64, 1) void qp_boxadmm_big_kernel(int B, int n, int m, const double* __restrict__ H, const double* __restrict__ h,
                                                               const double* __restrict__ A, const double* __restrict__ Alb,
                                                               const double* __restrict__ Aub, const double* __restrict__ xlb,
                                                               const double* __restrict__ xub, const double* __restrict__ x0,
                                                               const double* __restrict__ y0, pmpc_qp_settings s, double* __restrict__ Kws,
                                                               double* __restrict__ x, double* __restrict__ y, pmpc_qp_info* __restrict__ info) {
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int N = n + m;
    QpLds w;
    double* p = w.carve_xy(smem, n, m);
    double* rhsL = p; p += N;
    w.big_lds = p; p += BigKkt::LDS_DOUBLES;
    double* Wb = Kws + (size_t)b * (BigKkt::doubles(N) + ((QpLds::doubles_rest(n, m) + 1) & ~(size_t)1));   // (even stride: 16-byte panel reads)
    w.carve_rest_split(Wb + BigKkt::doubles(N), rhsL, n, m, Wb);
    pmpc_qp_info qi;
    boxadmm_solve<true>(w, n, m, H + (size_t)b * n * n, n, h + (size_t)b * n, A + (size_t)b * m * n, m, Alb + (size_t)b * m, Aub + (size_t)b * m,
                        xlb + (size_t)b * n, xub + (size_t)b * n, x0 ? x0 + (size_t)b * n : nullptr, y0 ? y0 + (size_t)b * N : nullptr, s, qi);
    const int ln = lane_id();
    for (int i = ln; i < n; i += WAVE) x[(size_t)b * n + i] = w.x[i];
    for (int i = ln; i < N; i += WAVE) y[(size_t)b * N + i] = w.y[i];
    if (ln == 0) info[b] = qi;
}

extern "C" size_t pmpc_internal_qp_big_ws_doubles(int n, int m) { return BigKkt::doubles(n + m) + ((QpLds::doubles_rest(n, m) + 1) & ~(size_t)1); }
extern "C" size_t pmpc_internal_qp_big_lds_bytes(int n, int m) {
    return (QpLds::doubles_xy(n, m) + (size_t)(n + m) + BigKkt::LDS_DOUBLES) * sizeof(double);
}
// launches on `stream` with the per-QP workspaces at Kws; 0 on success, -1 on a launch error
extern "C" int pmpc_internal_qp_big_launch(void* stream, double* Kws, int B, int n, int m, const double* H, const double* h, const double* A,
                                           const double* Alb, const double* Aub, const double* xlb, const double* xub, const double* x0,
                                           const double* y0, const pmpc_qp_settings* s, double* x, double* y, pmpc_qp_info* info) {
    const size_t lds = pmpc_internal_qp_big_lds_bytes(n, m);
    if (hipFuncSetAttribute((const void*)qp_boxadmm_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(qp_boxadmm_big_kernel, dim3(B), dim3(WAVE), lds, (hipStream_t)stream, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, *s, Kws,
                       x, y, info);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
