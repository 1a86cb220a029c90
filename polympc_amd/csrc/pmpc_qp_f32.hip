// polympc_amd — boxADMM<N, M, float>: the single-precision instantiation of the reference's QP solver (QPBase<..., Scalar = float>, qp_base.hpp:94-130;
// tests/solvers/qp/box_admm_test.cpp:85-115). Every quantity is a float as in the reference's templates: the settings of qp_solver_settings_t<float>,
// the `static constexpr scalar_t` constants, DIV_BY_ZERO_REGUL = regulariser<float>::value = 10e-5 (qp_base.hpp:84-86).
// One wavefront = one QP; the KKT matrix (full storage, column-major, (n+m)^2 floats) and the ADMM vectors live in LDS; LDL^T in a static order
// (K is symmetric quasi-definite for a positive semi-definite H), right-looking fma updates and column-oriented substitutions — operation for
// operation the float CPU restatement in its static order (the test suite checks the two bit for bit). A plain kernel: the fp64 paths are the tuned ones (the SQP solver computes in
// fp64 only, like every SQP test of the reference).
#include <hip/hip_runtime.h>
#include <cmath>
#include "pmpc_context.hpp"

namespace {

constexpr int WAVE = 64;
constexpr float F_RHO_MIN = 1e-6f, F_RHO_MAX = 1e+6f, F_RHO_EQ_FACTOR = 1e+3f;    // box_admm.hpp:56-59
constexpr float F_LOOSE_BOUNDS_THRESH = 1e+10f, F_EQ_TOL = 1e-4f;                  // qp_base.hpp:124-125
constexpr float F_DIV_BY_ZERO_REGUL = (float)10e-5;                                // qp_base.hpp:84-86

__device__ __forceinline__ int lane() { int l = threadIdx.x & (WAVE - 1); asm volatile("" : "+v"(l)); return l; }
__device__ __forceinline__ void wsync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ float wave_max(float v) {   // max is exact and order-free
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ int classify(float lb, float ub) {   // qp_base.hpp:195-222: 0 inequality, 1 equality, 2 loose
    if (lb < -F_LOOSE_BOUNDS_THRESH && ub > F_LOOSE_BOUNDS_THRESH) return 2;
    if (ub - lb < F_EQ_TOL) return 1;
    return 0;
}
__device__ __forceinline__ float rho_of(int type, float rho0) { return type == 2 ? F_RHO_MIN : (type == 1 ? F_RHO_EQ_FACTOR * rho0 : rho0); }

struct F32Lds {
    float *K, *x, *y, *xt, *q, *z, *zt, *zp, *rv, *rvi, *rb, *rbi, *rbp, *rhs, *lo, *hi, *hv, *kdg;
    int* type;
    __host__ __device__ static size_t floats(int n, int m) { const size_t N = (size_t)n + m; return N * N + 16 * N + 16; }
    __device__ void carve(float* p, int n, int m) {
        const int N = n + m;
        K = p; p += (size_t)N * N;
        x = p; p += n; y = p; p += N; xt = p; p += n; q = p; p += n; z = p; p += m; zt = p; p += m; zp = p; p += m;
        rv = p; p += m; rvi = p; p += m; rb = p; p += n; rbi = p; p += n; rbp = p; p += n; rhs = p; p += N;
        lo = p; p += N; hi = p; p += N; hv = p; p += n; kdg = p; p += n;
        type = (int*)p;   // N entries: [box (n) | general (m)]
    }
};

// static-order LDL^T in place (right-looking, fma updates; two rows per lane: N <= 128)
__device__ void factor_static(float* K, int N) {
    const int ln = lane();
    for (int k = 0; k < N; ++k) {
        const float dk = K[k + k * N];
        float col[2] = {0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = ln + WAVE * e;
            if (i > k && i < N) { col[e] = K[i + k * N]; K[i + k * N] = col[e] / dk; }
        }
        wsync();
        for (int j = k + 1; j < N; ++j) {
            const float ljk = K[j + k * N];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = ln + WAVE * e;
                if (i >= j && i < N) K[i + j * N] = fmaf(-col[e], ljk, K[i + j * N]);
            }
        }
        wsync();
    }
}

// construct_kkt_matrix (box_admm.hpp:209-223) / update_kkt_rho (:448-452), then the static-order LDL^T (right-looking, fma updates).
// The factorisation overwrites K in LDS, the reference keeps the unfactorised matrix and updates its diagonal in place: the primal diagonal is
// carried in kdg with exactly those updates (first: (H_ii + sigma) + rho_i; later: += rho_i - rho_i_prev) and K is rebuilt around it.
__device__ void build_and_factor(const F32Lds& w, int n, int m, const float* __restrict__ H, const float* __restrict__ A, float sigma, bool first) {
    const int N = n + m, ln = lane();
    for (int e = ln; e < N * N; e += WAVE) w.K[e] = 0.0f;
    wsync();
    for (int e = ln; e < n * n; e += WAVE) { const int i = e % n, j = e / n; w.K[i + j * N] = H[e]; }
    wsync();
    for (int i = ln; i < n; i += WAVE) {
        float d;
        if (first) { d = w.K[i + i * N]; d += sigma; d += w.rb[i]; }
        else { d = w.kdg[i]; d += (w.rb[i] - w.rbp[i]); }
        w.kdg[i] = d; w.K[i + i * N] = d;
    }
    for (int e = ln; e < m * n; e += WAVE) { const int i = e % m, j = e / m; w.K[(n + i) + j * N] = A[e]; }
    for (int i = ln; i < m; i += WAVE) w.K[(n + i) + (n + i) * N] = -w.rvi[i];
    wsync();
    factor_static(w.K, N);
}

// column-oriented substitutions in the order of the static-order CPU restatement: in place on v (N floats in LDS)
__device__ void ldlt_solve(const float* K, int N, float* v) {
    const int ln = lane();
    for (int j = 0; j < N; ++j) {
        const float xj = v[j];
        for (int i = j + 1 + ln; i < N; i += WAVE) v[i] = fmaf(-K[i + j * N], xj, v[i]);
        wsync();
    }
    for (int i = ln; i < N; i += WAVE) v[i] = v[i] / K[i + i * N];
    wsync();
    for (int j = N - 1; j >= 0; --j) {
        const float xj = v[j];
        for (int i = ln; i < j; i += WAVE) v[i] = fmaf(-K[j + i * N], xj, v[i]);
        wsync();
    }
}

__global__ __launch_bounds__(64) void qp_boxadmm_f32_kernel(int B, int n, int m, const float* __restrict__ Hb, const float* __restrict__ hb,
                                                            const float* __restrict__ Ab, const float* __restrict__ Albb, const float* __restrict__ Aubb,
                                                            const float* __restrict__ xlbb, const float* __restrict__ xubb, const float* __restrict__ x0b,
                                                            const float* __restrict__ y0b, pmpc_qp_settings sd, float* __restrict__ xo, float* __restrict__ yo,
                                                            pmpc_qp_info* __restrict__ info) {
    extern __shared__ float smem_f[];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int N = n + m, ln = lane();
    F32Lds w; w.carve(smem_f, n, m);
    const float* H = Hb + (size_t)b * n * n; const float* h = hb + (size_t)b * n; const float* A = Ab + (size_t)b * m * n;
    // qp_solver_settings_t<float>
    const float eps_rel = (float)sd.eps_rel, eps_abs = (float)sd.eps_abs, sigma = (float)sd.sigma, alpha = (float)sd.alpha, tol = (float)sd.adaptive_rho_tolerance;
    float rho = (float)sd.rho;
    for (int i = ln; i < n; i += WAVE) {
        w.hv[i] = h[i]; w.lo[i] = xlbb[(size_t)b * n + i]; w.hi[i] = xubb[(size_t)b * n + i];
        w.x[i] = x0b ? x0b[(size_t)b * n + i] : 0.0f; w.q[i] = w.x[i];
        w.type[i] = classify(w.lo[i], w.hi[i]);
    }
    for (int i = ln; i < m; i += WAVE) {
        w.lo[n + i] = Albb[(size_t)b * m + i]; w.hi[n + i] = Aubb[(size_t)b * m + i];
        w.type[n + i] = classify(w.lo[n + i], w.hi[n + i]);
    }
    for (int i = ln; i < N; i += WAVE) w.y[i] = y0b ? y0b[(size_t)b * N + i] : 0.0f;
    wsync();
    for (int i = ln; i < m; i += WAVE) { float a = 0.0f; for (int j = 0; j < n; ++j) a += A[i + j * m] * w.x[j]; w.z[i] = a; }   // z = A x_guess
    auto rho_vec_update = [&](float rho0) {   // box_admm.hpp:357-396
        for (int i = ln; i < m; i += WAVE) { w.rv[i] = rho_of(w.type[n + i], rho0); w.rvi[i] = 1.0f / w.rv[i]; }
        for (int i = ln; i < n; i += WAVE) { w.rb[i] = rho_of(w.type[i], rho0); w.rbi[i] = 1.0f / w.rb[i]; }
        wsync();
    };
    int rho_updates = 1;
    rho_vec_update(rho);
    build_and_factor(w, n, m, H, A, sigma, true);
    int status = PMPC_QP_UNSOLVED, iter = 1;
    float max_Ax_z = 0.0f, max_Hx = 0.0f, res_prim = 1.0f, res_dual = 1.0f, rho_estimate = 0.0f;
    for (; iter <= sd.max_iter; ++iter) {
        for (int i = ln; i < m; i += WAVE) w.zp[i] = w.z[i];
        for (int i = ln; i < n; i += WAVE) w.rhs[i] = ((sigma * w.x[i] - w.hv[i]) + w.rb[i] * w.q[i]) - w.y[m + i];   // compute_kkt_rhs :351-355
        for (int i = ln; i < m; i += WAVE) w.rhs[n + i] = w.z[i] - w.rvi[i] * w.y[i];
        wsync();
        ldlt_solve(w.K, N, w.rhs);
        for (int i = ln; i < n; i += WAVE) {
            const float xt = w.rhs[i];
            float xx = alpha * xt; xx += (1 - alpha) * xx;   // quirk Q1 (:129-130)
            w.x[i] = xx;
            float qq = xx + w.rbi[i] * w.y[m + i];
            qq = fminf(fmaxf(qq, w.lo[i]), w.hi[i]);
            w.q[i] = qq;
            w.y[m + i] += w.rb[i] * (xx - qq);
        }
        for (int i = ln; i < m; i += WAVE) {
            const float zt = w.zp[i] + w.rvi[i] * (w.rhs[n + i] - w.y[i]);
            float zz = alpha * zt;
            zz += (1 - alpha) * w.zp[i] + w.rvi[i] * w.y[i];
            zz = fminf(fmaxf(zz, w.lo[n + i]), w.hi[n + i]);
            w.z[i] = zz;
            w.y[i] += w.rv[i] * ((alpha * zt + (1 - alpha) * w.zp[i]) - zz);
        }
        wsync();
        const bool check = sd.check_termination != 0 && iter % sd.check_termination == 0;
        const bool adapt = sd.adaptive_rho && iter % sd.adaptive_rho_interval == 0;
        if (check || adapt) {   // residuals_update :398-415
            float nAx = 0.0f, nz = 0.0f, nx = 0.0f, nHx = 0.0f, nATy = 0.0f, nh = 0.0f, nyb = 0.0f, rp = 0.0f, rq = 0.0f, rd = 0.0f;
            for (int i = ln; i < m; i += WAVE) {
                float a = 0.0f; for (int j = 0; j < n; ++j) a += A[i + j * m] * w.x[j];
                nAx = fmaxf(nAx, fabsf(a)); nz = fmaxf(nz, fabsf(w.z[i])); rp = fmaxf(rp, fabsf(a - w.z[i]));
            }
            for (int i = ln; i < n; i += WAVE) {
                float hx = 0.0f; for (int j = 0; j < n; ++j) hx += H[i + j * n] * w.x[j];
                float aty = 0.0f; for (int k = 0; k < m; ++k) aty += A[k + i * m] * w.y[k];
                nx = fmaxf(nx, fabsf(w.x[i])); nHx = fmaxf(nHx, fabsf(hx)); nATy = fmaxf(nATy, fabsf(aty));
                nh = fmaxf(nh, fabsf(w.hv[i])); nyb = fmaxf(nyb, fabsf(w.y[m + i]));
                rq = fmaxf(rq, fabsf(w.x[i] - w.q[i]));
                rd = fmaxf(rd, fabsf(((hx + w.hv[i]) + aty) + w.y[m + i]));
            }
            max_Ax_z = wave_max(fmaxf(nAx, fmaxf(nz, nx)));
            max_Hx = wave_max(fmaxf(nHx, fmaxf(nATy, fmaxf(nh, nyb))));
            res_prim = wave_max(rp) + wave_max(rq);
            res_dual = wave_max(rd);
        }
        if (check && res_prim <= eps_abs + eps_rel * max_Ax_z && res_dual <= eps_abs + eps_rel * max_Hx) { status = PMPC_QP_SOLVED; break; }
        if (adapt) {
            const float rpn = res_prim / (max_Ax_z + F_DIV_BY_ZERO_REGUL);
            const float rdn = res_dual / (max_Hx + F_DIV_BY_ZERO_REGUL);
            float new_rho = rho * sqrtf(rpn / (rdn + F_DIV_BY_ZERO_REGUL));
            new_rho = fmaxf(F_RHO_MIN, fminf(new_rho, F_RHO_MAX));
            rho_estimate = new_rho;
            if (new_rho < rho / tol || new_rho > rho * tol) {
                for (int i = ln; i < n; i += WAVE) w.rbp[i] = w.rb[i];
                wsync();
                rho = new_rho;
                rho_vec_update(rho);
                ++rho_updates;
                build_and_factor(w, n, m, H, A, sigma, false);
            }
        }
    }
    if (iter > sd.max_iter) status = PMPC_QP_MAX_ITER_EXCEEDED;
    for (int i = ln; i < n; i += WAVE) xo[(size_t)b * n + i] = w.x[i];
    for (int i = ln; i < N; i += WAVE) yo[(size_t)b * N + i] = w.y[i];
    bool bad = false;
    for (int i = ln; i < n; i += WAVE) bad |= (w.x[i] - w.x[i]) != 0.0f;
    for (int i = ln; i < N; i += WAVE) bad |= (w.y[i] - w.y[i]) != 0.0f;
    const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (ln == 0) {
        pmpc_qp_info qi; qi.status = status; qi.iter = iter; qi.rho_updates = rho_updates; qi.flags = anybad ? PMPC_FLAG_NONFINITE : 0;
        qi.rho_estimate = rho_estimate; qi.res_prim = res_prim; qi.res_dual = res_dual;
        info[b] = qi;
    }
}

// ADMM<N, M, float> (admm.hpp:112-212, OSQP form; tests/solvers/qp/admm_solver_test.cpp:84-113): the box constraints stacked under the general ones,
// z / y / rho over the ME = m + n rows of A_e = [A ; I], one (2n + m)-row KKT matrix [H + sigma I, A_e' ; A_e, -diag(1/rho)] (construct_kkt_matrix
// :249-263) in LDS, no quirk Q1. A rho update changes the lower diagonal only (update_kkt_rho :490-494): K is rebuilt (the same values) and re-factorised.
struct F32AdmmLds {
    float *K, *x, *y, *z, *zp, *rv, *rvi, *rhs, *lo, *hi, *hv;
    int* type;
    __host__ __device__ static size_t floats(int n, int m) { const size_t NK = 2 * (size_t)n + m; return NK * NK + 12 * NK + 16; }
    __device__ void carve(float* p, int n, int m) {
        const int ME = n + m, NK = n + ME;
        K = p; p += (size_t)NK * NK;
        x = p; p += n; y = p; p += ME; z = p; p += ME; zp = p; p += ME; rv = p; p += ME; rvi = p; p += ME; rhs = p; p += NK;
        lo = p; p += ME; hi = p; p += ME; hv = p; p += n;
        type = (int*)p;
    }
};

__device__ void admm_build_and_factor(const F32AdmmLds& w, int n, int m, const float* __restrict__ H, const float* __restrict__ A, float sigma) {
    const int ME = n + m, NK = n + ME, ln = lane();
    for (int e = ln; e < NK * NK; e += WAVE) w.K[e] = 0.0f;
    wsync();
    for (int e = ln; e < n * n; e += WAVE) { const int i = e % n, j = e / n; w.K[i + j * NK] = H[e]; }
    wsync();
    for (int i = ln; i < n; i += WAVE) { w.K[i + i * NK] += sigma; w.K[(n + m + i) + i * NK] = 1.0f; }
    for (int e = ln; e < m * n; e += WAVE) { const int i = e % m, j = e / m; w.K[(n + i) + j * NK] = A[e]; }
    for (int i = ln; i < ME; i += WAVE) w.K[(n + i) + (n + i) * NK] = -1.0f * w.rvi[i];
    wsync();
    factor_static(w.K, NK);
}

__global__ __launch_bounds__(64) void qp_admm_f32_kernel(int B, int n, int m, const float* __restrict__ Hb, const float* __restrict__ hb,
                                                         const float* __restrict__ Ab, const float* __restrict__ Albb, const float* __restrict__ Aubb,
                                                         const float* __restrict__ xlbb, const float* __restrict__ xubb, const float* __restrict__ x0b,
                                                         const float* __restrict__ y0b, pmpc_qp_settings sd, float* __restrict__ xo, float* __restrict__ yo,
                                                         pmpc_qp_info* __restrict__ info) {
    extern __shared__ float smem_f[];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int ME = n + m, NK = n + ME, ln = lane();
    F32AdmmLds w; w.carve(smem_f, n, m);
    const float* H = Hb + (size_t)b * n * n; const float* h = hb + (size_t)b * n; const float* A = Ab + (size_t)b * m * n;
    const float eps_rel = (float)sd.eps_rel, eps_abs = (float)sd.eps_abs, sigma = (float)sd.sigma, alpha = (float)sd.alpha, tol = (float)sd.adaptive_rho_tolerance;
    float rho = (float)sd.rho;
    for (int i = ln; i < n; i += WAVE) {
        w.hv[i] = h[i]; w.lo[m + i] = xlbb[(size_t)b * n + i]; w.hi[m + i] = xubb[(size_t)b * n + i];
        w.x[i] = x0b ? x0b[(size_t)b * n + i] : 0.0f;
        w.type[m + i] = classify(w.lo[m + i], w.hi[m + i]);
    }
    for (int i = ln; i < m; i += WAVE) {
        w.lo[i] = Albb[(size_t)b * m + i]; w.hi[i] = Aubb[(size_t)b * m + i];
        w.type[i] = classify(w.lo[i], w.hi[i]);
    }
    for (int i = ln; i < ME; i += WAVE) w.y[i] = y0b ? y0b[(size_t)b * ME + i] : 0.0f;
    wsync();
    for (int i = ln; i < m; i += WAVE) { float a = 0.0f; for (int j = 0; j < n; ++j) a += A[i + j * m] * w.x[j]; w.z[i] = a; }   // z = A_e x_guess
    for (int i = ln; i < n; i += WAVE) w.z[m + i] = w.x[i];
    auto rho_vec_update = [&](float rho0) {   // admm.hpp:405-440
        for (int i = ln; i < ME; i += WAVE) { w.rv[i] = rho_of(w.type[i], rho0); w.rvi[i] = 1.0f / w.rv[i]; }
        wsync();
    };
    int rho_updates = 1;
    rho_vec_update(rho);
    admm_build_and_factor(w, n, m, H, A, sigma);
    int status = PMPC_QP_UNSOLVED, iter = 1;
    float max_Ax_z = 0.0f, max_Hx = 0.0f, res_prim = 1.0f, res_dual = 1.0f, rho_estimate = 0.0f;
    for (; iter <= sd.max_iter; ++iter) {
        for (int i = ln; i < ME; i += WAVE) w.zp[i] = w.z[i];
        for (int i = ln; i < n; i += WAVE) w.rhs[i] = sigma * w.x[i] - w.hv[i];   // compute_kkt_rhs :390-394
        for (int i = ln; i < ME; i += WAVE) w.rhs[n + i] = w.z[i] - w.rvi[i] * w.y[i];
        wsync();
        ldlt_solve(w.K, NK, w.rhs);
        for (int i = ln; i < n; i += WAVE) w.x[i] = alpha * w.rhs[i] + (1 - alpha) * w.x[i];   // :152
        for (int i = ln; i < ME; i += WAVE) {
            const float zt = w.zp[i] + w.rvi[i] * (w.rhs[n + i] - w.y[i]);
            float zz = alpha * zt;
            zz += (1 - alpha) * w.zp[i] + w.rvi[i] * w.y[i];
            zz = fminf(fmaxf(zz, w.lo[i]), w.hi[i]);
            w.z[i] = zz;
            w.y[i] += w.rv[i] * ((alpha * zt + (1 - alpha) * w.zp[i]) - zz);
        }
        wsync();
        const bool check = sd.check_termination != 0 && iter % sd.check_termination == 0;
        const bool adapt = sd.adaptive_rho && iter % sd.adaptive_rho_interval == 0;
        if (check || adapt) {   // residuals_update :442-462
            float nAx = 0.0f, nz = 0.0f, nx = 0.0f, nHx = 0.0f, nATy = 0.0f, nh = 0.0f, nyb = 0.0f, rp = 0.0f, rb = 0.0f, rd = 0.0f;
            for (int i = ln; i < m; i += WAVE) {
                float a = 0.0f; for (int j = 0; j < n; ++j) a += A[i + j * m] * w.x[j];
                nAx = fmaxf(nAx, fabsf(a)); rp = fmaxf(rp, fabsf(a - w.z[i]));
            }
            for (int i = ln; i < ME; i += WAVE) nz = fmaxf(nz, fabsf(w.z[i]));
            for (int i = ln; i < n; i += WAVE) {
                float hx = 0.0f; for (int j = 0; j < n; ++j) hx += H[i + j * n] * w.x[j];
                float aty = 0.0f; for (int k = 0; k < m; ++k) aty += A[k + i * m] * w.y[k];
                nx = fmaxf(nx, fabsf(w.x[i])); nHx = fmaxf(nHx, fabsf(hx)); nATy = fmaxf(nATy, fabsf(aty));
                nh = fmaxf(nh, fabsf(w.hv[i])); nyb = fmaxf(nyb, fabsf(w.y[m + i]));
                rb = fmaxf(rb, fabsf(w.x[i] - w.z[m + i]));
                rd = fmaxf(rd, fabsf(((hx + w.hv[i]) + aty) + w.y[m + i]));
            }
            max_Ax_z = wave_max(fmaxf(fmaxf(nAx, nx), nz));
            max_Hx = wave_max(fmaxf(nHx, fmaxf(nATy, fmaxf(nh, nyb))));
            res_prim = fmaxf(wave_max(rp), wave_max(rb));
            res_dual = wave_max(rd);
        }
        if (check && res_prim <= eps_abs + eps_rel * max_Ax_z && res_dual <= eps_abs + eps_rel * max_Hx) { status = PMPC_QP_SOLVED; break; }
        if (adapt) {
            const float rpn = res_prim / (max_Ax_z + F_DIV_BY_ZERO_REGUL);
            const float rdn = res_dual / (max_Hx + F_DIV_BY_ZERO_REGUL);
            float new_rho = rho * sqrtf(rpn / (rdn + F_DIV_BY_ZERO_REGUL));
            new_rho = fmaxf(F_RHO_MIN, fminf(new_rho, F_RHO_MAX));
            rho_estimate = new_rho;
            if (new_rho < rho / tol || new_rho > rho * tol) {
                rho = new_rho;
                rho_vec_update(rho);
                ++rho_updates;
                admm_build_and_factor(w, n, m, H, A, sigma);
            }
        }
    }
    if (iter > sd.max_iter) status = PMPC_QP_MAX_ITER_EXCEEDED;
    for (int i = ln; i < n; i += WAVE) xo[(size_t)b * n + i] = w.x[i];
    for (int i = ln; i < ME; i += WAVE) yo[(size_t)b * ME + i] = w.y[i];
    bool bad = false;
    for (int i = ln; i < n; i += WAVE) bad |= (w.x[i] - w.x[i]) != 0.0f;
    for (int i = ln; i < ME; i += WAVE) bad |= (w.y[i] - w.y[i]) != 0.0f;
    const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (ln == 0) {
        pmpc_qp_info qi; qi.status = status; qi.iter = iter; qi.rho_updates = rho_updates; qi.flags = anybad ? PMPC_FLAG_NONFINITE : 0;
        qi.rho_estimate = rho_estimate; qi.res_prim = res_prim; qi.res_dual = res_dual;
        info[b] = qi;
    }
}

}  // namespace

extern "C" {

static pmpc_status f32_solve_dev(bool osqp, pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                 const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                 const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !H || !h || !xlb || !xub || !settings || !x || !y || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (m > 0 && (!A || !Alb || !Aub)) return PMPC_ERR_INVALID_ARGUMENT;
    if ((x0 == nullptr) != (y0 == nullptr)) return PMPC_ERR_INVALID_ARGUMENT;
    if (settings->linear_solver != 0) return PMPC_ERR_INVALID_ARGUMENT;   // the static order only
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t lds = (osqp ? F32AdmmLds::floats(n, m) : F32Lds::floats(n, m)) * sizeof(float);
    if ((osqp ? 2 * n + m : n + m) > 2 * WAVE || lds > ctx->lds_limit) return PMPC_ERR_UNSUPPORTED_SIZE;   // two KKT rows per lane, the matrix in LDS
    auto kern = osqp ? qp_admm_f32_kernel : qp_boxadmm_f32_kernel;
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PMPC_POISON_DEVICE(ctx);
    hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), lds, ctx->stream, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, *settings, x, y, info);
    HIPCHK(hipGetLastError());
    return PMPC_OK;
}
pmpc_status pmpc_qp_boxadmm_solve_batch_f32_dev(pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                                const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                                const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info) {
    return f32_solve_dev(false, ctx, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, settings, x, y, info);
}
pmpc_status pmpc_qp_admm_solve_batch_f32_dev(pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                             const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                             const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info) {
    return f32_solve_dev(true, ctx, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, settings, x, y, info);
}

static pmpc_status f32_solve_host(bool osqp, pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                  const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                  const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info) {
    if (!ctx || B < 0 || n < 1 || m < 0 || !H || !h || !xlb || !xub || !settings || !x || !y || !info) return PMPC_ERR_INVALID_ARGUMENT;
    if (m > 0 && (!A || !Alb || !Aub)) return PMPC_ERR_INVALID_ARGUMENT;
    if ((x0 == nullptr) != (y0 == nullptr)) return PMPC_ERR_INVALID_ARGUMENT;
    if (B == 0) return PMPC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t Bz = (size_t)B;
    const void* host[9] = {H, h, A, Alb, Aub, xlb, xub, x0, y0};
    const size_t count[9] = {Bz * n * n, Bz * n, Bz * m * n, Bz * m, Bz * m, Bz * n, Bz * n, Bz * n, Bz * (n + m)};
    float* dev[9] = {nullptr};
    for (int k = 0; k < 9; ++k) {
        if (!host[k] || count[k] == 0) continue;
        void* p = nullptr;
        const pmpc_status st = ensure_scratch(ctx, k, count[k] * sizeof(float), &p);
        if (st != PMPC_OK) return st;
        HIPCHK(hipMemcpyAsync(p, host[k], count[k] * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        dev[k] = (float*)p;
    }
    void *dx = nullptr, *dy = nullptr, *di = nullptr;
    pmpc_status st = ensure_scratch(ctx, 9, Bz * n * sizeof(float), &dx); if (st != PMPC_OK) return st;
    st = ensure_scratch(ctx, 10, Bz * (n + m) * sizeof(float), &dy); if (st != PMPC_OK) return st;
    st = ensure_scratch(ctx, 11, Bz * sizeof(pmpc_qp_info), &di); if (st != PMPC_OK) return st;
    if (m == 0) { dev[2] = dev[3] = dev[4] = (float*)dx; }   // never read (m = 0), but the device entry wants non-null pointers only when m > 0
    st = f32_solve_dev(osqp, ctx, B, n, m, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6], dev[7], dev[8], settings,
                       (float*)dx, (float*)dy, (pmpc_qp_info*)di);
    if (st != PMPC_OK) return st;
    HIPCHK(hipMemcpyAsync(x, dx, Bz * n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(y, dy, Bz * (n + m) * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(info, di, Bz * sizeof(pmpc_qp_info), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PMPC_OK;
}
pmpc_status pmpc_qp_boxadmm_solve_batch_f32(pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                            const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                            const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info) {
    return f32_solve_host(false, ctx, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, settings, x, y, info);
}
pmpc_status pmpc_qp_admm_solve_batch_f32(pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                         const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                         const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info) {
    return f32_solve_host(true, ctx, B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, settings, x, y, info);
}

}  // extern "C"
