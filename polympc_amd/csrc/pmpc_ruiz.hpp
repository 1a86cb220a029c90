// polympc_amd — Ruiz equilibration of a dense QP on the device, one wavefront per QP instance, in place.
//
// Replaces polympc::RuizEquilibration<Scalar, N, M, DENSE> (/root/reference/src/solvers/qp_preconditioners.hpp): compute
// :160-233, scale :352-357, unscale(x, y) :359-364, unscale(H, h, A, Al, Au, l, u) :367-383; call site in the SQP loop
// sqp_base.hpp:605-611 / :661-665. Same sweep count, loop condition (the norm measured BEFORE a sweep decides whether the
// next one runs), singularity guard, cost scaling and coefficient-wise association order:
//   D.asDiagonal() * H * D.asDiagonal()                 ->  (D_i * H_ij) * D_j
//   (1/c) * Dinv.asDiagonal() * H * Dinv.asDiagonal()   ->  (((1/c) * Dinv_i) * H_ij) * Dinv_j
// sqrt and the divisions are the correctly rounded IEEE operations, so the scaled problem equals the CPU restatement's
// bit for bit.
//
// Mapping: lane i owns row i of H and A for the row norms and the scaling passes (loads coalesced down the columns); the
// column norms walk column `lane` (strided across lanes, served by L2). D, E and the per-sweep factors live in LDS or
// global scratch supplied by the caller (n + m doubles each).
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_qp.hpp"

namespace pmpc {

struct RuizScratch {
    double* D;    // n   accumulated column scaling
    double* E;    // m   accumulated row scaling of A
    double* mD;   // n   factors of the current sweep
    double* mE;   // m
};

// Strided walks with eight loads in flight (a loop with one load -> use -> store per trip waits a full memory round trip per element; every
// element below is independent of the others — maxima are order-free, the scalings act entry by entry — so batching changes no result).
constexpr int RUIZ_CH = 8;
// max_k |M[base + k*stride]|, k < cnt
__device__ __forceinline__ double ruiz_absmax(const double* M, size_t base, size_t stride, int cnt) {
    double r = 0.0;
    for (int k0 = 0; k0 < cnt; k0 += RUIZ_CH) {
        double t[RUIZ_CH];
#pragma unroll
        for (int u = 0; u < RUIZ_CH; ++u) t[u] = M[base + (size_t)((k0 + u < cnt) ? k0 + u : cnt - 1) * stride];
#pragma unroll
        for (int u = 0; u < RUIZ_CH; ++u) r = fmax(r, fabs(t[u]));
    }
    return r;
}
// M[base + k*stride] <- (a * M[...]) * f(k), k < cnt   (f: per-column factor from `col`, or its reciprocal)
template <bool RECIP>
__device__ __forceinline__ void ruiz_scale_row(double* M, size_t base, size_t stride, int cnt, double a, const double* col) {
    for (int k0 = 0; k0 < cnt; k0 += RUIZ_CH) {
        double t[RUIZ_CH], c[RUIZ_CH];
#pragma unroll
        for (int u = 0; u < RUIZ_CH; ++u) { const int k = (k0 + u < cnt) ? k0 + u : cnt - 1; t[u] = M[base + (size_t)k * stride]; c[u] = col[k]; }
#pragma unroll
        for (int u = 0; u < RUIZ_CH; ++u) if (k0 + u < cnt) M[base + (size_t)(k0 + u) * stride] = (a * t[u]) * (RECIP ? 1.0 / c[u] : c[u]);
    }
}
// M[base + k*stride] *= g, k < cnt
__device__ __forceinline__ void ruiz_mul_row(double* M, size_t base, size_t stride, int cnt, double g) {
    for (int k0 = 0; k0 < cnt; k0 += RUIZ_CH) {
        double t[RUIZ_CH];
#pragma unroll
        for (int u = 0; u < RUIZ_CH; ++u) t[u] = M[base + (size_t)((k0 + u < cnt) ? k0 + u : cnt - 1) * stride];
#pragma unroll
        for (int u = 0; u < RUIZ_CH; ++u) if (k0 + u < cnt) M[base + (size_t)(k0 + u) * stride] = t[u] * g;
    }
}

// H(i,j) = H[i + j*ldh], A(i,j) = A[i + j*lda]. Returns the cost scaling c.
__device__ __noinline__ double ruiz_compute_wave(int n, int m, double* H, int ldh, double* h, double* A, int lda, double* Al, double* Au,
                                          double* l, double* u, const RuizScratch& w) {
    const int ln = lane_id();
    constexpr int max_iter = 4;
    constexpr double approx_zero = 2.220446049250313e-16, tolerance = 1e-3;
    double c = 1.0;
    for (int k = ln; k < n; k += WAVE) w.D[k] = 1.0;
    for (int k = ln; k < m; k += WAVE) w.E[k] = 1.0;
    wsync();
    double scaling_norm = 10 * tolerance;
    for (int iter = 0; iter < max_iter && (1.0 - scaling_norm) >= tolerance; ++iter) {
        double mx = 0.0;
        for (int i = ln; i < m; i += WAVE) {   // row norms of A
            const double r = ruiz_absmax(A, i, lda, n);
            mx = fmax(mx, r);
            w.mE[i] = (r < approx_zero) ? 1.0 : r;
        }
        for (int j = ln; j < n; j += WAVE) {   // column norms of [H ; A]
            const double r = fmax(ruiz_absmax(H, (size_t)j * ldh, 1, n), m > 0 ? ruiz_absmax(A, (size_t)j * lda, 1, m) : 0.0);
            mx = fmax(mx, r);
            w.mD[j] = (r < approx_zero) ? 1.0 : r;
        }
        scaling_norm = wave_max(mx);
        for (int k = ln; k < n; k += WAVE) w.mD[k] = 1.0 / ::sqrt(w.mD[k]);
        for (int k = ln; k < m; k += WAVE) w.mE[k] = 1.0 / ::sqrt(w.mE[k]);
        wsync();
        for (int i = ln; i < n; i += WAVE) {
            const double di = w.mD[i];
            ruiz_scale_row<false>(H, i, ldh, n, di, w.mD);
            h[i] = h[i] * di;
            w.D[i] = w.D[i] * di;
        }
        for (int i = ln; i < m; i += WAVE) {
            const double ei = w.mE[i];
            ruiz_scale_row<false>(A, i, lda, n, ei, w.mD);
            w.E[i] = w.E[i] * ei;
        }
        wfence();
        wsync();
        // cost scaling: gamma = 1 / max(mean of the column norms of H, |h|_inf)
        double hi = 0.0;
        for (int j = ln; j < n; j += WAVE) {
            const double r = ruiz_absmax(H, (size_t)j * ldh, 1, n);
            w.mD[j] = r;
            hi = fmax(hi, fabs(h[j]));
        }
        double h_inf = wave_max(hi);
        h_inf = h_inf > approx_zero ? h_inf : 1.0;
        wsync();
        double sum = 0.0;
        for (int j = 0; j < n; ++j) sum += w.mD[j];   // every lane, same order
        const double gamma = 1.0 / fmax(sum / n, h_inf);
        for (int i = ln; i < n; i += WAVE) {
            ruiz_mul_row(H, i, ldh, n, gamma);
            h[i] *= gamma;
        }
        c *= gamma;
        wfence();
        wsync();
    }
    for (int k = ln; k < m; k += WAVE) { Au[k] = Au[k] * w.E[k]; Al[k] = Al[k] * w.E[k]; }
    for (int k = ln; k < n; k += WAVE) { const double di = 1.0 / w.D[k]; l[k] = l[k] * di; u[k] = u[k] * di; }
    wsync();
    return c;
}

// x <- x.*D ; y_A <- (1/c) (y_A.*E) ; y_box <- (1/c) (y_box./D)
__device__ __noinline__ void ruiz_unscale_solution_wave(int n, int m, const double* D, const double* E, double c, double* x, double* y) {
    const int ln = lane_id();
    const double ic = 1 / c;
    for (int k = ln; k < n; k += WAVE) { x[k] = x[k] * D[k]; y[m + k] = ic * (y[m + k] * (1.0 / D[k])); }
    for (int k = ln; k < m; k += WAVE) y[k] = ic * (y[k] * E[k]);
    wsync();
}

// back to the unscaled problem data (the SQP keeps using H, h, A, bounds after the QP)
__device__ __noinline__ void ruiz_unscale_problem_wave(int n, int m, double* H, int ldh, double* h, double* A, int lda, double* Al, double* Au,
                                                 double* l, double* u, const double* D, const double* E, double c) {
    const int ln = lane_id();
    const double ic = 1 / c;
    for (int i = ln; i < n; i += WAVE) {
        const double di = ic * (1.0 / D[i]);
        ruiz_scale_row<true>(H, i, ldh, n, di, D);
    }
    for (int i = ln; i < m; i += WAVE) {
        const double ei = 1.0 / E[i];
        ruiz_scale_row<true>(A, i, lda, n, ei, D);
        Au[i] = Au[i] * ei; Al[i] = Al[i] * ei;
    }
    for (int k = ln; k < n; k += WAVE) { h[k] = ic * (h[k] * (1.0 / D[k])); l[k] = l[k] * D[k]; u[k] = u[k] * D[k]; }
    wfence();
    wsync();
}

// ---- batched kernels behind pmpc_qp_ruiz_*_batch: one wavefront per QP, problem data in place in global memory
static __global__ __launch_bounds__(64) void ruiz_compute_kernel(int B, int n, int m, double* H, double* h, double* A, double* Alb, double* Aub,
                                                          double* xlb, double* xub, double* D, double* E, double* c, double* scratch) {
    const int b = blockIdx.x;
    if (b >= B) return;
    RuizScratch w{D + (size_t)b * n, E + (size_t)b * m, scratch + (size_t)b * (n + m), scratch + (size_t)b * (n + m) + n};
    const double cb = ruiz_compute_wave(n, m, H + (size_t)b * n * n, n, h + (size_t)b * n, A + (size_t)b * m * n, m, Alb + (size_t)b * m,
                                        Aub + (size_t)b * m, xlb + (size_t)b * n, xub + (size_t)b * n, w);
    if (lane_id() == 0) c[b] = cb;
}
static __global__ __launch_bounds__(64) void ruiz_unscale_solution_kernel(int B, int n, int m, const double* D, const double* E, const double* c,
                                                                   double* x, double* y) {
    const int b = blockIdx.x;
    if (b >= B) return;
    ruiz_unscale_solution_wave(n, m, D + (size_t)b * n, E + (size_t)b * m, c[b], x + (size_t)b * n, y + (size_t)b * (n + m));
}

}  // namespace pmpc
