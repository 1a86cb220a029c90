// polympc_amd — wave-cooperative dense box-ADMM QP solve (device side), one 64-lane wavefront per QP instance.
//
// Replaces boxADMM::solve_impl and its helpers (/root/reference/src/solvers/box_admm.hpp:88-223,336-452) together with
// the Eigen::LDLT factor/solve it calls (src/utils/helpers.hpp:38-43). Same update order, constants and quirks
// (Q1: x = alpha*x_tilde; x += (1-alpha)*x — box_admm.hpp:129-130).
//
// Layout: the lower triangle of the (n+m)x(n+m) KKT matrix lives in LDS, packed by columns (element (i,j), i>=j, at
// j*N - j*(j+1)/2 + i: column walks are contiguous across lanes); lane i owns KKT row i (rows i, i+64, ... when n+m > 64). The LDL^T uses the static
// elimination order (no pivoting — K is symmetric quasi-definite), right-looking, fused
// multiply-add on the trailing update and the substitutions. H and A stay in HBM/L2 and are streamed only when the
// residuals are evaluated (every check_termination-th iteration).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/polympc_amd.h"

namespace pmpc {

constexpr int WAVE = 64;
#ifndef PMPC_BIG_MEM_BATCH
#define PMPC_BIG_MEM_BATCH 32
#endif
constexpr int BIG_MEM_BATCH = PMPC_BIG_MEM_BATCH;   // loads in flight per lane in the row walks of the HBM-factor kernels (BFGS products and update, H x of the residuals): one memory round trip per batch
constexpr double RHO_MIN = 1e-6, RHO_MAX = 1e+6, RHO_EQ_FACTOR = 1e+3;   // box_admm.hpp:56-59
constexpr double LOOSE_BOUNDS_THRESH = 1e+10, EQ_TOL = 1e-4;            // qp_base.hpp:124-125
constexpr double DIV_BY_ZERO_REGUL = 10e-10;                            // qp_base.hpp:79-82
constexpr double PMPC_SCHUR_COND_GATE = 1e7;   // the block-structured kernel's gate on max S_ii max |(S^-1)_ii|, S = 1/rho + A Q A' (pmpc_qp_schur.hpp): below it the range-space solve follows exact arithmetic as closely as the reference's pivoted LDL^T does (robot / CSTR / parking streams at rho0 = 0.1 .. 1e3, tests/test_oracle_pins.py); the error then grows like the estimate squared (2e-5 at 4e7, 1e-3 at 4e9, nothing at 4e10)
constexpr double PMPC_COND_GATE = 1e10;   // conditioning gate of the kernels that eliminate the constraint block first (PMPC_FLAG_ILLCOND, include/polympc_amd.h): max_i S_ii / min_k |pivot_k|

// opaque on purpose: per-lane index arithmetic derived from it is recomputed where it is used instead of being hoisted
// to the kernel prologue and kept (or spilled) for the whole SQP loop
__device__ __forceinline__ int lane_id() { int l = threadIdx.x & (WAVE - 1); asm volatile("" : "+v"(l)); return l; }
// One wavefront per workgroup: every producer / consumer pair of LDS or workspace data sits in the SAME wavefront, whose
// memory instructions are issued and performed in program order. Synchronisation is therefore a wavefront-scope fence
// (orders the compiler, emits no s_waitcnt vmcnt(0) / s_barrier — a workgroup-scope __syncthreads would stall every
// phase boundary until all outstanding workspace stores have been acknowledged by L2).
__device__ __forceinline__ void wfence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
__device__ __forceinline__ void wsync() { wfence(); __builtin_amdgcn_wave_barrier(); }

// A zero the optimiser cannot see through. Added to the per-lane element offset of global-memory accesses inside the SQP /
// ADMM loops, it keeps their address arithmetic next to the access: otherwise every one of the O(n) distinct 64-bit
// addresses is computed once outside the loops (they are loop-invariant) and then lives in — or is spilled from — a
// register pair for the whole kernel.
__device__ __forceinline__ unsigned opaque_zero() { unsigned z = 0; asm volatile("" : "+v"(z)); return z; }

// max over the 64 lanes, returned to every lane. DPP reduction (quad swaps, row mirrors, row broadcasts) — six VALU steps
// without an LDS round trip (a __shfl_xor butterfly costs two ds_bpermute per step); the total ends up in lane 63 and is
// broadcast through an SGPR pair. max is exact, so the combination order does not matter.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return fmax(v, __hiloint2double(hi, lo));
}
__device__ __forceinline__ double wave_max(double v) {
    v = dpp_max_step<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_max_step<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_max_step<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_max_step<0x140, 0xf>(v);   // row_mirror: every lane of a 16-lane row holds the row maximum
    v = dpp_max_step<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3: lane 63 holds the wave maximum
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, returned to every lane, through the same six DPP steps: the partial sums are added PAIRWISE OVER ADJACENT LANES, level by
// level (lanes 2i and 2i+1, then quads, half rows, rows, row pairs, the two halves) — a fixed balanced tree, which the CPU restatement of the callers
// walks (every addition is commutative, so which side a partner arrives on does not matter). Rows that a step's row mask leaves out add +0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_step(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add_step<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add_step<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add_step<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add_step<0x140, 0xf>(v);   // row_mirror: every lane of a 16-lane row holds the row sum
    v = dpp_add_step<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
    v = dpp_add_step<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3: lane 63 holds (row 3 + row 2) + (row 1 + row 0)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// a / b for a wave-uniform divisor, correctly rounded like the IEEE division it replaces, at 5 VALU operations per
// quotient instead of the ~14 of the generic expansion: y = RN(1/b) is computed once (a true division); then
//   q0 = RN(a*y), r0 = a - b*q0 (fma), q1 = RN(q0 + r0*y)   -> q1 is a faithful quotient
//   r1 = a - b*q1 (exact, fma), q = RN(q1 + r1*y)            -> RN(a/b) (Markstein's theorem for a correctly rounded y)
// ok(): the divisor lies in a conservative exponent window (callers branch ONCE on it, wave-uniformly, and use the
// generic division otherwise).
struct UniformDiv {
    double b, y;
    __device__ __forceinline__ explicit UniformDiv(double b_) : b(b_), y(1.0 / b_) {}
    __device__ __forceinline__ bool ok() const { const double ab = fabs(b); return __builtin_amdgcn_readfirstlane((int)(ab > 1e-150 && ab < 1e150)) != 0; }
    __device__ __forceinline__ double operator()(double a) const {
        const double q0 = a * y;
        const double q1 = fma(fma(-q0, b, a), y, q0);
        return fma(fma(-q1, b, a), y, q1);
    }
};

// sequential (index-ordered) sum of an LDS vector, evaluated redundantly by every lane (broadcast reads): the
// same association order as the reference's scalar loops, no cross-lane traffic
// (loads in chunks of 8 independent reads, then the ordered add chain: a loop with one load per iteration waits a full
// LDS round trip per element)
__device__ __forceinline__ double seq_sum(const double* v, int n) {
    double a = 0.0;
    for (int i0 = 0; i0 < n; i0 += 8) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[(i0 + j < n) ? i0 + j : 0];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (i0 + j < n) a += t[j];
    }
    return a;
}
__device__ __forceinline__ double seq_dot(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int i0 = 0; i0 < n; i0 += 8) {
        double ta[8], tb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int i = (i0 + j < n) ? i0 + j : 0; ta[j] = a[i]; tb[j] = b[i]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (i0 + j < n) s += ta[j] * tb[j];
    }
    return s;
}
__device__ __forceinline__ double lds_inf_norm(const double* v, int n) {
    double r = 0.0;
    for (int i = lane_id(); i < n; i += WAVE) r = fmax(r, fabs(v[i]));
    return wave_max(r);
}

// qp_base.hpp:195-222 ; 0 inequality, 1 equality, 2 loose
__device__ __forceinline__ int classify_bounds(double lb, double ub) {
    if (lb < -LOOSE_BOUNDS_THRESH && ub > LOOSE_BOUNDS_THRESH) return 2;
    if (ub - lb < EQ_TOL) return 1;
    return 0;
}
__device__ __forceinline__ double rho_of(int type, double rho0) {  // box_admm.hpp:357-396
    return type == 2 ? RHO_MIN : (type == 1 ? RHO_EQ_FACTOR * rho0 : rho0);
}

// LDS work area of one QP instance
struct QpLds {
    static constexpr int WAVE_DUMMY = 64;
    double* K; int N;              // packed lower triangle of the (n+m)x(n+m) factor, by columns
    __host__ __device__ static int koff(int N_, int j) { return j * N_ - (j * (j + 1)) / 2; }
    __device__ __forceinline__ int off(int j) const { return j * N - (j * (j + 1)) / 2; }
    double *x, *y, *z, *q, *zt, *zprev, *rho, *rhoinv, *rhob, *rhobinv, *kdiag, *rhs, *t1, *t2;
    double* trp = nullptr;         // transpositions of the pivoted LDL^T (linear_solver = 1), N entries stored as doubles; LDS-resident mode only
    double* big_lds = nullptr;     // large-instance mode (pmpc_qp_big.hpp): BigKkt::LDS_DOUBLES doubles of LDS; K then points at the tile workspace in HBM
    __host__ __device__ static size_t kdoubles(int N_) { return (size_t)N_ * (N_ + 1) / 2; }
    __host__ __device__ static size_t doubles(int n, int m) {
        const int N = n + m;
        return kdoubles(N) + 2 * WAVE_DUMMY /*per-lane dummy slots behind K*/ + 3 * (size_t)n /*x q kdiag(n part)*/ + (size_t)N /*y*/ + 5 * (size_t)m /*z zt zprev rho rhoinv*/ +
               2 * (size_t)n /*rhob rhobinv*/ + (size_t)N /*rhs*/ + 2 * (size_t)N /*t1 t2*/ + (size_t)m /*kdiag m part*/ + (size_t)N /*trp*/ + 8;
    }
    // register-resident QP path: only the result vectors live in LDS
    __host__ __device__ static size_t doubles_xy(int n, int m) { return 2 * (size_t)n + (size_t)m + 8; }
    __device__ __forceinline__ double* carve_xy(double* base, int n, int m) {
        N = n + m; K = nullptr;
        double* p = base;
        x = p; p += n; y = p; p += N;
        q = z = zt = zprev = rho = rhoinv = rhob = rhobinv = kdiag = rhs = t1 = t2 = nullptr;
        return p;
    }
    // large-instance mode: x, y as carve_xy; K in an HBM workspace; every other vector in a caller-provided LDS region
    __host__ __device__ static size_t doubles_rest(int n, int m) { return 3 * (size_t)n + 5 * (size_t)m + 4 * (size_t)(n + m); }
    __device__ __forceinline__ void carve_rest(double* p, int n, int m, double* K_hbm) {
        N = n + m; K = K_hbm;
        q = p; p += n; kdiag = p; p += N;
        z = p; p += m; zt = p; p += m; zprev = p; p += m; rho = p; p += m; rhoinv = p; p += m;
        rhob = p; p += n; rhobinv = p; p += n; rhs = p; p += N; t1 = p; p += N; t2 = p; p += N;
    }
    // large-instance mode, round 2: as carve_rest with the right-hand side of the substitutions in LDS and the other vectors wherever `p` points (HBM)
    __device__ __forceinline__ void carve_rest_split(double* p, double* rhs_lds, int n, int m, double* K_hbm) {
        N = n + m; K = K_hbm;
        q = p; p += n; kdiag = p; p += N;
        z = p; p += m; zt = p; p += m; zprev = p; p += m; rho = p; p += m; rhoinv = p; p += m;
        rhob = p; p += n; rhobinv = p; p += n; rhs = rhs_lds; t1 = p; p += N; t2 = p; p += N;
    }
    __device__ __forceinline__ double* carve(double* base, int n, int m) {
        N = n + m;
        double* p = base;
        K = p; p += kdoubles(N) + 2 * WAVE_DUMMY;   // K[kdoubles(N) + lane], K[kdoubles(N) + 64 + lane]: dummy slots of the branch-free factor
        x = p; p += n; q = p; p += n; kdiag = p; p += N; y = p; p += N;
        z = p; p += m; zt = p; p += m; zprev = p; p += m; rho = p; p += m; rhoinv = p; p += m;
        rhob = p; p += n; rhobinv = p; p += n; rhs = p; p += N; t1 = p; p += N; t2 = p; p += N; trp = p; p += N;
        return p;
    }
};

// K (lower triangle) <- [H + diag(kdiag[0:n]) ; A, diag(kdiag[n:])]  (construct_kkt_matrix, box_admm.hpp:209-223)
__device__ __forceinline__ void kkt_build(const QpLds& w, int n, int m, const double* __restrict__ H, int ldh, const double* __restrict__ A, int lda) {
    const int ln = lane_id();
    const int N = n + m;
    if (N <= 2 * WAVE) {
        // at most two rows per lane (i0 = lane, i1 = lane + 64): the global loads of eight columns are issued together before the
        // first LDS store, instead of one dependent load -> store round trip per column segment
        const int i0 = ln, i1 = ln + WAVE;
        const bool h0 = i0 < N, h1 = i1 < N;
        constexpr int CB = 8;
        for (int j0 = 0; j0 < n; j0 += CB) {
            double e0[CB], e1[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int j = (j0 + u < n) ? j0 + u : n - 1;
                e0[u] = 0.0; e1[u] = 0.0;
                if (h0 && i0 > j) e0[u] = (i0 < n) ? H[(size_t)j * ldh + i0] : A[(size_t)j * lda + (i0 - n)];
                if (h1 && i1 > j) e1[u] = (i1 < n) ? H[(size_t)j * ldh + i1] : A[(size_t)j * lda + (i1 - n)];
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int j = j0 + u;
                if (j < n) {
                    const int o = w.off(j);
                    if (h0 && i0 >= j) w.K[o + i0] = (i0 == j) ? w.kdiag[i0] : e0[u];
                    if (h1 && i1 >= j) w.K[o + i1] = (i1 == j) ? w.kdiag[i1] : e1[u];
                }
            }
        }
        for (int j = 0; j < m; ++j) {
            const int o = w.off(n + j);
            for (int i = j + ln; i < m; i += WAVE) w.K[o + n + i] = (i == j) ? w.kdiag[n + i] : 0.0;
        }
        wsync();
        return;
    }
    for (int j = 0; j < n; ++j) {
        const int o = w.off(j);
        for (int i = j + ln; i < n; i += WAVE) w.K[o + i] = (i == j) ? w.kdiag[i] : H[(size_t)j * ldh + i];
        for (int r = ln; r < m; r += WAVE) w.K[o + n + r] = A[(size_t)j * lda + r];
    }
    for (int j = 0; j < m; ++j) {
        const int o = w.off(n + j);
        for (int i = j + ln; i < m; i += WAVE) w.K[o + n + i] = (i == j) ? w.kdiag[n + i] : 0.0;
    }
    wsync();
}

// wave-uniform lane broadcast of a double (v_readlane with a scalar lane index: no LDS crossbar round trip)
__device__ __forceinline__ double bcast_uniform(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Right-looking LDL^T for systems of at most 128 rows: two rows per lane (i0 = lane, i1 = lane + 64). The unscaled column stays in two
// registers and the multiplier l_jk comes out of the scaled registers with v_readlane (no LDS round trip through the column just
// written); the trailing update takes four columns at a time — loads issued together, then the fma, then the stores. Like the
// substitutions this loop is bound by its instruction count (one wavefront per SIMD): the entry index is a running value (one add per
// column, the second row is + 64), and a lane without an entry in a column is redirected to its private dummy slot behind the packed
// triangle with one select — branch-free, and an index, not a pointer, is selected (compiler hazard 7 in DESIGN.md).
template <bool TWO>
__device__ __forceinline__ void kkt_factor_rows(const QpLds& w, int N) {
    const int ln = lane_id();
    double* K = w.K;
    const int i0 = ln, i1 = ln + WAVE;
    const bool h0 = i0 < N, h1 = TWO && i1 < N;
    const int d0 = (int)QpLds::kdoubles(N) + ln, d1 = d0 + QpLds::WAVE_DUMMY;
    const int NL = N < WAVE ? N : WAVE;
    constexpr int FC = 4;                              // columns per batch of loads (8 costs the same per column and lengthens the remainder)
    int ok = 0;                                        // off(k)
    for (int k = 0; k < N; ++k) {
        const double dk = K[ok + k];
        const bool a0 = h0 && i0 > k, a1 = h1 && i1 > k;
        const int p0 = a0 ? ok + i0 : d0, p1 = a1 ? ok + i1 : d1;
        const double c0 = K[p0];
        const double c1 = TWO ? K[p1] : 0.0;
        const double s0 = c0 / dk, s1 = TWO ? c1 / dk : 0.0;   // scaled column: l_jk for row j lives in lane j (s0) / lane j - 64 (s1)
        K[p0] = s0;
        if (TWO) K[p1] = s1;
        ok += N - 1 - k;                               // off(k + 1)
        int ij = ok + i0;                              // index of (i0, j) for the running column j
        int inc = N - 2 - k;                           // off(j + 1) - off(j)
        int j = k + 1;
        for (; j + FC <= NL; j += FC) {                  // columns below 64: rows lane from j on, rows lane + 64 always
            double l[FC], e0[FC], e1[FC]; int q0[FC], q1[FC];
#pragma unroll
            for (int u = 0; u < FC; ++u) {
                l[u] = bcast_uniform(s0, j + u);
                q0[u] = (a0 && i0 >= j + u) ? ij : d0;
                if (TWO) q1[u] = a1 ? ij + WAVE : d1;
                ij += inc; --inc;
            }
#pragma unroll
            for (int u = 0; u < FC; ++u) { e0[u] = K[q0[u]]; if (TWO) e1[u] = K[q1[u]]; }
#pragma unroll
            for (int u = 0; u < FC; ++u) { K[q0[u]] = fma(-c0, l[u], e0[u]); if (TWO) K[q1[u]] = fma(-c1, l[u], e1[u]); }
        }
        for (; j < NL; ++j) {
            const double l = bcast_uniform(s0, j);
            const int q0 = (a0 && i0 >= j) ? ij : d0, q1 = a1 ? ij + WAVE : d1;
            ij += inc; --inc;
            const double e0 = K[q0], e1 = TWO ? K[q1] : 0.0;
            K[q0] = fma(-c0, l, e0);
            if (TWO) K[q1] = fma(-c1, l, e1);
        }
        if (TWO) {                                     // columns from 64 on: only rows lane + 64, from j on
            for (; j + FC <= N; j += FC) {
                double l[FC], e1[FC]; int q1[FC];
#pragma unroll
                for (int u = 0; u < FC; ++u) {
                    l[u] = bcast_uniform(s1, j + u - WAVE);
                    q1[u] = (a1 && i1 >= j + u) ? ij + WAVE : d1;
                    ij += inc; --inc;
                }
#pragma unroll
                for (int u = 0; u < FC; ++u) e1[u] = K[q1[u]];
#pragma unroll
                for (int u = 0; u < FC; ++u) K[q1[u]] = fma(-c1, l[u], e1[u]);
            }
            for (; j < N; ++j) {
                const double l = bcast_uniform(s1, j - WAVE);
                const int q1 = (a1 && i1 >= j) ? ij + WAVE : d1;
                ij += inc; --inc;
                const double e1 = K[q1];
                K[q1] = fma(-c1, l, e1);
            }
        }
        wsync();
    }
}

// in-place LDL^T, static order, right-looking (factorise_kkt_matrix, box_admm.hpp:336-341)
__device__ __forceinline__ void kkt_factor(const QpLds& w, int N) {
    const int ln = lane_id();
    double* K = w.K;
    if (N <= 2 * WAVE) {
        if (N <= WAVE) kkt_factor_rows<false>(w, N); else kkt_factor_rows<true>(w, N);
        return;
    }
    for (int k = 0; k < N; ++k) {
        const int ok = w.off(k);
        const double dk = K[ok + k];
        // scale column k, keep the unscaled entries in t1
        for (int i = k + 1 + ln; i < N; i += WAVE) {
            const double c = K[ok + i];
            w.t1[i] = c;
            K[ok + i] = c / dk;
        }
        wsync();
        for (int j = k + 1; j < N; ++j) {
            const double ljk = K[ok + j];
            const int oj = w.off(j);
            for (int i = j + ln; i < N; i += WAVE) K[oj + i] = fma(-w.t1[i], ljk, K[oj + i]);
        }
        wsync();
    }
}

// Substitutions for systems of at most 128 rows: two rows per lane, held in registers (c0: row lane, c1: row lane + 64). A single
// wavefront per SIMD issues one instruction every four cycles whatever its kind, so these loops are bound by their instruction COUNT:
//  - the factor entry of a step is read through a running LDS pointer (one add per step; the second row is the same address + 64
//    entries, an immediate offset), never clamped: a lane without an entry in the column reads a neighbouring entry or the dummy slots
//    behind the triangle, and its result is dropped;
//  - only the row set that really shrinks during a phase is masked (rows `lane` while the pivots are below 64, rows `lane + 64`
//    above); the other row of the lane is either always inside the column or belongs to a lane whose registers are never read;
//  - the pivot value comes out of the registers with v_readlane; the loads of eight steps are issued ahead of the (serial) fma chain.
// Same operations on every entry, in the same order, as the generic loops below (forward: columns ascending, backward: descending).
template <bool TWO>
__device__ __forceinline__ void kkt_solve_rows(const QpLds& w, int N, double* v) {
    const int ln = lane_id();
    const double* K = w.K;
    const int i0 = ln, i1 = ln + WAVE;
    const bool h0 = i0 < N, h1 = TWO && i1 < N;
    const int r0 = h0 ? i0 : 0, r1 = h1 ? i1 : 0;
    double c0 = h0 ? v[i0] : 0.0, c1 = h1 ? v[i1] : 0.0;
    const int nA = (N - 1 < WAVE) ? N - 1 : WAVE;   // forward steps whose pivot lives in c0
    // ---- forward, pivots in c0: L(i, j) at off(j) + i
    const double* pa = K + i0;
    int inc = N - 1;                                  // off(j + 1) - off(j)
    int j = 0;
    for (; j + 8 <= nA; j += 8) {
        double f0[8], f1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { f0[u] = pa[0]; if (TWO) f1[u] = pa[WAVE]; pa += inc; --inc; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double xj = bcast_uniform(c0, j + u);
            const double t0 = fma(-f0[u], xj, c0);
            if (TWO) c1 = fma(-f1[u], xj, c1);
            c0 = (i0 > j + u) ? t0 : c0;
        }
    }
    for (; j < nA; ++j) {
        const double f0 = pa[0], f1 = TWO ? pa[WAVE] : 0.0;
        pa += inc; --inc;
        const double xj = bcast_uniform(c0, j);
        const double t0 = fma(-f0, xj, c0);
        if (TWO) c1 = fma(-f1, xj, c1);
        c0 = (i0 > j) ? t0 : c0;
    }
    if (TWO) {   // ---- forward, pivots in c1 (j from 64 on): only rows lane + 64 are still below them
        for (; j + 8 <= N - 1; j += 8) {
            double f1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { f1[u] = pa[WAVE]; pa += inc; --inc; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double xj = bcast_uniform(c1, j + u - WAVE);
                const double t1 = fma(-f1[u], xj, c1);
                c1 = (i1 > j + u) ? t1 : c1;
            }
        }
        for (; j < N - 1; ++j) {
            const double f1 = pa[WAVE];
            pa += inc; --inc;
            const double xj = bcast_uniform(c1, j - WAVE);
            const double t1 = fma(-f1, xj, c1);
            c1 = (i1 > j) ? t1 : c1;
        }
    }
    // ---- diagonal
    const int o0 = w.off(r0), o1 = w.off(r1);
    c0 = c0 / K[o0 + r0];
    if (TWO) c1 = c1 / K[o1 + r1];
    // ---- backward: L(j, i) at off(i) + j, pivots descending
    j = N - 1;
    if (TWO) {   // pivots in c1 (j >= 64): every row below 64, and the rows lane + 64 above the pivot
        const double* pb0 = K + o0 + j;
        const double* pb1 = K + o1 + j;
        for (; j - 8 >= WAVE - 1; j -= 8) {
            double f0[8], f1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { f0[u] = pb0[-u]; f1[u] = pb1[-u]; }
            pb0 -= 8; pb1 -= 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double xj = bcast_uniform(c1, j - u - WAVE);
                c0 = fma(-f0[u], xj, c0);
                const double t1 = fma(-f1[u], xj, c1);
                c1 = (i1 < j - u) ? t1 : c1;
            }
        }
        for (; j >= WAVE; --j) {
            const double f0 = pb0[0], f1 = pb1[0];
            --pb0; --pb1;
            const double xj = bcast_uniform(c1, j - WAVE);
            c0 = fma(-f0, xj, c0);
            const double t1 = fma(-f1, xj, c1);
            c1 = (i1 < j) ? t1 : c1;
        }
    }
    {   // pivots in c0 (j below 64): rows lane < j only
        const double* pb0 = K + o0 + j;
        for (; j - 8 >= 0; j -= 8) {
            double f0[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) f0[u] = pb0[-u];
            pb0 -= 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double xj = bcast_uniform(c0, j - u);
                const double t0 = fma(-f0[u], xj, c0);
                c0 = (i0 < j - u) ? t0 : c0;
            }
        }
        for (; j > 0; --j) {
            const double f0 = pb0[0];
            --pb0;
            const double xj = bcast_uniform(c0, j);
            const double t0 = fma(-f0, xj, c0);
            c0 = (i0 < j) ? t0 : c0;
        }
    }
    wsync();
    if (h0) v[i0] = c0;
    if (h1) v[i1] = c1;
    wsync();
}

// v <- K^{-1} v  (linear_solver.solve, box_admm.hpp:123), v in LDS
__device__ __forceinline__ void kkt_solve(const QpLds& w, int N, double* v) {
    const int ln = lane_id();
    const double* K = w.K;
    if (N <= 2 * WAVE) {
        if (N <= WAVE) kkt_solve_rows<false>(w, N, v); else kkt_solve_rows<true>(w, N, v);
        return;
    }
    for (int j = 0; j < N - 1; ++j) {
        const double xj = v[j];
        for (int i = j + 1 + ln; i < N; i += WAVE) v[i] = fma(-K[w.off(j) + i], xj, v[i]);
        wsync();
    }
    for (int i = ln; i < N; i += WAVE) v[i] = v[i] / K[w.off(i) + i];
    wsync();
    for (int j = N - 1; j > 0; --j) {
        const double xj = v[j];
        for (int i = ln; i < j; i += WAVE) v[i] = fma(-K[w.off(i) + j], xj, v[i]);
        wsync();
    }
}

// sum_j M[j * sj + i * si] * vec[j], one add chain with j ascending (the order of the CPU restatement); eight global loads and eight LDS
// reads are in flight before the chain consumes them: full chunks without any clamping, then one clamped chunk for the remainder
template <int CH = 8>
__device__ __forceinline__ double seq_dot_strided(const double* __restrict__ M, size_t sj, size_t si, int i, int cnt, const double* vec) {
    double a = 0.0;
    const double* __restrict__ p = M + (size_t)i * si;
    int j0 = 0;
    for (; j0 + CH <= cnt; j0 += CH) {
        double e[CH], x[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) { e[u] = p[(size_t)u * sj]; x[u] = vec[j0 + u]; }
        p += (size_t)CH * sj;
#pragma unroll
        for (int u = 0; u < CH; ++u) a += e[u] * x[u];
    }
    if (j0 < cnt) {
        const int rem = cnt - j0;
        double e[CH], x[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) { const int uu = (u < rem) ? u : rem - 1; e[u] = p[(size_t)uu * sj]; x[u] = vec[j0 + uu]; }
#pragma unroll
        for (int u = 0; u < CH; ++u) if (u < rem) a += e[u] * x[u];
    }
    return a;
}

struct QpResidualState { double max_Ax_z_norm, max_Hx_ATy_h_norm, res_prim, res_dual; };

// residuals_update, box_admm.hpp:398-415 (H, A streamed from global memory, coalesced down the columns)
__device__ __forceinline__ void qp_residuals(const QpLds& w, int n, int m, const double* __restrict__ H, int ldh, const double* __restrict__ h,
                                    const double* __restrict__ A, int lda, QpResidualState& r) {
    const int ln = lane_id();
    // The three products keep one add chain per row (columns ascending: the order of the CPU restatement); what is batched is the
    // loads — eight global loads and eight LDS reads are in flight before the chain consumes them (one dependent L2 round trip per
    // term made this routine cost four substitutions).
    auto dot_col = [&](const double* __restrict__ M, int ld, int i, int cnt, const double* vec) -> double { return seq_dot_strided(M, (size_t)ld, 1, i, cnt, vec); };
    auto dot_row = [&](const double* __restrict__ M, int ld, int i, int cnt, const double* vec) -> double { return seq_dot_strided(M, 1, (size_t)ld, i, cnt, vec); };
    double nAx = 0, nz = 0, nx = 0, rp = 0;
    for (int i = ln; i < m; i += WAVE) {
        const double a = dot_col(A, lda, i, n, w.x);
        nAx = fmax(nAx, fabs(a)); nz = fmax(nz, fabs(w.z[i])); rp = fmax(rp, fabs(a - w.z[i]));
    }
    double nHx = 0, nATy = 0, nh = 0, nyb = 0, rq = 0, rd = 0;
    for (int i = ln; i < n; i += WAVE) {
        const double a = dot_col(H, ldh, i, n, w.x);
        const double b = (m > 0) ? dot_row(A, lda, i, m, w.y) : 0.0;
        nx = fmax(nx, fabs(w.x[i])); nHx = fmax(nHx, fabs(a)); nATy = fmax(nATy, fabs(b));
        nh = fmax(nh, fabs(h[i])); nyb = fmax(nyb, fabs(w.y[m + i]));
        rq = fmax(rq, fabs(w.x[i] - w.q[i]));
        rd = fmax(rd, fabs(((a + h[i]) + b) + w.y[m + i]));
    }
    // max is exact and order-free: the norms that are only ever used through their maximum are merged before the reductions
    nAx = wave_max(fmax(nAx, fmax(nz, nx))); nz = nAx; nx = nAx;
    nHx = wave_max(fmax(fmax(nHx, nATy), fmax(nh, nyb))); nATy = nHx; nh = nHx; nyb = nHx;
    rp = wave_max(rp); rq = wave_max(rq); rd = wave_max(rd);
    r.max_Ax_z_norm = fmax(nAx, fmax(nz, nx));
    r.max_Hx_ATy_h_norm = fmax(nHx, fmax(nATy, fmax(nh, nyb)));
    r.res_prim = rp + rq;
    r.res_dual = rd;
}

__device__ __forceinline__ void rho_vec_update(const QpLds& w, int n, int m, const double* Alb, const double* Aub, const double* xlb,
                                      const double* xub, double rho0) {
    const int ln = lane_id();
    for (int i = ln; i < m; i += WAVE) { const double r = rho_of(classify_bounds(Alb[i], Aub[i]), rho0); w.rho[i] = r; w.rhoinv[i] = 1.0 / r; }
    for (int i = ln; i < n; i += WAVE) { const double r = rho_of(classify_bounds(xlb[i], xub[i]), rho0); w.rhob[i] = r; w.rhobinv[i] = 1.0 / r; }
}

// ---- linear_solver = 1: LDL^T with Eigen::LDLT's diagonal pivoting, operation for operation the CPU restatement's Eigen-style policy
// (SURVEY Appendix B): at step k the largest remaining |diagonal| (first occurrence) is swapped to position k; left-looking update
//   temp_j = d_j l_kj,  a_kk -= sum_j l_kj temp_j,  a_ik -= sum_j l_ij temp_j  (products and adds, j ascending — no fma),  l_ik = a_ik / a_kk  (skipped for a zero pivot);
// solve: transpositions, forward substitution (x_i -= l_ij x_j, j ascending), D^+ (entries with |d| <= 1/DBL_MAX give 0), backward substitution
// (x_i -= l_ji x_j, j ascending from i+1: a serial chain per row — evaluated by every lane on broadcast LDS reads), transpositions back.
// Packed lower triangle in LDS as for the static order. Slow by construction (dot-product form, serial backward pass): a policy for
// indefinite Hessians and for cross-checks, not a fast path.
__device__ __forceinline__ void kkt_factor_pivoted(const QpLds& w, int N) {
    const int ln = lane_id();
    double* K = w.K; double* temp = w.t1; double* tr = w.trp;
    auto at = [&](int i, int j) -> double& { return K[w.off(j) + i]; };   // i >= j
    auto swp = [](double& a, double& b) { const double t = a; a = b; b = t; };
    for (int k = 0; k < N; ++k) {
        // largest remaining |diagonal|, first occurrence (a NaN at position k keeps k, as the scalar loop does)
        const double dkk = fabs(at(k, k));
        double bv = -1.0; int bi = N;
        for (int i = k + ln; i < N; i += WAVE) { const double v = fabs(at(i, i)); if (v > bv) { bv = v; bi = i; } }
        const double mx = wave_max(bv);
        int big = -(int)wave_max((bv == mx && bi < N) ? -(double)bi : -1.0e9);
        if (dkk != dkk) big = k;
        big = __builtin_amdgcn_readfirstlane(big);
        if (ln == 0) tr[k] = (double)big;
        if (big != k) {
            for (int j = ln; j < k; j += WAVE) swp(at(k, j), at(big, j));
            for (int i = big + 1 + ln; i < N; i += WAVE) swp(at(i, k), at(i, big));
            if (ln == 0) swp(at(k, k), at(big, big));
            for (int i = k + 1 + ln; i < big; i += WAVE) swp(at(i, k), at(big, i));
        }
        wsync();
        if (k > 0) {
            for (int j = ln; j < k; j += WAVE) temp[j] = at(j, j) * at(k, j);
            wsync();
            double acc = 0.0;
            { int o = 0; for (int j = 0; j < k; ++j) { acc += K[o + k] * temp[j]; o += N - 1 - j; } }
            for (int i = k + 1 + ln; i < N; i += WAVE) {
                double a = 0.0;
                int o = 0;
                for (int j0 = 0; j0 < k; j0 += 8) {
                    double e[8], t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int j = (j0 + u < k) ? j0 + u : k - 1; e[u] = K[w.off(j) + i]; t[u] = temp[j]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (j0 + u < k) a += e[u] * t[u];
                }
                (void)o;
                at(i, k) -= a;
            }
            if (ln == 0) at(k, k) -= acc;
            wsync();
        }
        const double akk = at(k, k);
        const bool valid = fabs(akk) > 0.0;
        if (k == 0 && !valid) { for (int j = ln; j < N; j += WAVE) tr[j] = (double)j; wsync(); return; }
        if (valid) for (int i = k + 1 + ln; i < N; i += WAVE) at(i, k) = at(i, k) / akk;
        wsync();
    }
}
__device__ __forceinline__ void kkt_solve_pivoted(const QpLds& w, int N, double* v) {
    const int ln = lane_id();
    const double* K = w.K; const double* tr = w.trp;
    if (ln == 0) for (int k = 0; k < N; ++k) { const int t = (int)tr[k]; if (t != k) { const double a = v[k]; v[k] = v[t]; v[t] = a; } }
    wsync();
    for (int j = 0; j < N - 1; ++j) {
        const double xj = v[j];
        const int o = w.off(j);
        for (int i = j + 1 + ln; i < N; i += WAVE) v[i] = v[i] - K[o + i] * xj;
        wsync();
    }
    const double tol = 1.0 / 1.7976931348623157e308;
    for (int i = ln; i < N; i += WAVE) { const double d = K[w.off(i) + i]; v[i] = (fabs(d) > tol) ? v[i] / d : 0.0; }
    wsync();
    for (int i = N - 2; i >= 0; --i) {   // row i: a serial chain over j = i+1 .. N-1 (every lane evaluates it on broadcast reads; lane 0 stores)
        double a = v[i];
        const int o = w.off(i);
        for (int j0 = i + 1; j0 < N; j0 += 8) {
            double e[8], x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = (j0 + u < N) ? j0 + u : N - 1; e[u] = K[o + j]; x[u] = v[j]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (j0 + u < N) a -= e[u] * x[u];
        }
        wsync();
        if (ln == 0) v[i] = a;
        wsync();
    }
    if (ln == 0) for (int k = N - 1; k >= 0; --k) { const int t = (int)tr[k]; if (t != k) { const double a = v[k]; v[k] = v[t]; v[t] = a; } }
    wsync();
}

// large-instance linear algebra (pmpc_qp_big.hpp, included after this header by its users)
__device__ __forceinline__ void big_build(double* W, int n, int m, const double* __restrict__ H, int ldh, const double* __restrict__ A, int lda, const double* kdiag);
// The two row-parallel loops of the damped BFGS update on the dense workspace (SqpDevice::bfgs_update / rank2_update_rows), rows i = 64 w + lane with stride
// 64 NW: wavefront w of a team of NW (BigTeam). Row i sees exactly the operations it saw on one wavefront.
template <int NW, int MEMCH_>
__device__ __forceinline__ void bfgs_rows_products(const double* Hw, int ldw, int n, const double* step, const double* lgn, const double* lg, double* Bs, double* y, int w) {
    for (int i = lane_id() + WAVE * w; i < n; i += WAVE * NW) {
        Bs[i] = seq_dot_strided<MEMCH_>(Hw, (size_t)ldw, 1, i, n, step);   // row i of B times s: one add chain, columns ascending
        y[i] = lgn[i] - lg[i];
    }
}
template <int NW, int CH, bool FAST>
__device__ __forceinline__ void bfgs_rows_rank2(double* Hw, int ldw, int n, const double* Bs, const double* r, double sBs, double sr, int w) {
    const UniformDiv by_sBs(sBs), by_sr(sr);
    for (int i = lane_id() + WAVE * w; i < n; i += WAVE * NW) {
        const double Bsi = Bs[i], ri = r[i];
        double* __restrict__ row = Hw + i;
        for (int j0 = 0; j0 < n; j0 += CH) {
            double b[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) b[u] = row[(size_t)((j0 + u < n) ? j0 + u : n - 1) * ldw];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (j0 + u < n) {
                    double t = b[u];
                    if constexpr (FAST) { t += by_sBs(-Bsi * Bs[j0 + u]); t += by_sr(ri * r[j0 + u]); }
                    else { t += (-Bsi * Bs[j0 + u]) / sBs; t += (ri * r[j0 + u]) / sr; }
                    row[(size_t)(j0 + u) * ldw] = t;
                }
        }
    }
}

// The wavefronts that work on ONE instance's linear algebra (round 5). NW = 1: the one-wavefront-per-instance kernels — every function below is then the
// code it was. NW = 4 (sqp_kernel<..., WG4>: one workgroup of four wavefronts per instance, small batches — a lone instance's latency is what a
// receding-horizon controller waits for): wavefront 0 runs the instance's serial code (linearisation, BFGS, ADMM vector updates, line search) and posts the
// three heavy routines — condensed build + blocked factorisation, the condensed solve — to a mailbox in LDS (BigMail, big_helper_loop); all four
// wavefronts then execute the routine with the independent pieces dealt round-robin (tiles of the build, tile-row groups of the left-looking updates,
// 64-row slots of the forward pass, column quadruples of the backward pass, 64-entry chunks of the two sparse products). Every fma chain stays on ONE
// wavefront in the order it had, so the results are bit-identical to the one-wavefront kernels (and to PIVOT_CONDENSED). sync() is a workgroup barrier
// with workgroup-scope memory ordering (the pieces exchange data through the HBM factor workspace and LDS).
template <int NW>
struct BigTeam {
    int w = 0;   // this wavefront's index in the team
    __device__ __forceinline__ void sync() const { if constexpr (NW > 1) __syncthreads(); else { wfence(); wsync(); } }
    // barrier for exchanges through LDS only (the triangular passes, the sparse products: the factor is read-only there): no wait for outstanding global loads, so
    // that panel entries requested for the NEXT block column stay in flight across it (__syncthreads() drains vmcnt and with it every prefetch)
    __device__ __forceinline__ void sync_lds() const { if constexpr (NW > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else { wfence(); wsync(); } }
    __device__ __forceinline__ bool mine(int item) const { if constexpr (NW > 1) return (item % NW) == w; else return true; }
    __device__ __forceinline__ bool lead() const { if constexpr (NW > 1) return w == 0; else return true; }
};
enum { BIG_OP_EXIT = 0, BIG_OP_FACTOR = 1, BIG_OP_SOLVE = 2, BIG_OP_STAGE2 = 3, BIG_OP_STAGE1 = 4, BIG_OP_BFGS_BS = 5, BIG_OP_BFGS_R2 = 6 };   // mailbox of a team (BigMail, pmpc_qp_big.hpp)
template <class JV> struct BigMail;
template <int NW> __device__ __forceinline__ double big_factor(double* W, int N, double* dl, const BigTeam<NW>& team);
template <bool SLIM, int NW> __device__ __forceinline__ void big_solve(const double* W, int N, double* v, double* bx, const BigTeam<NW>& team);
template <class JV, int NW> __device__ __forceinline__ double big_build_condensed(double* W, int n, int m, const double* __restrict__ H, int ldh, const double* kdiag,
                                                                                  const double* rho, const JV& jv, const BigTeam<NW>& team, double* red4);
template <bool SLIM, int NW, class JV> __device__ __forceinline__ void big_cond_solve(const double* K, int n, int m, double* rhs, const double* rho, double* scr, const JV& jv,
                                                                                      const BigTeam<NW>& team, long long* t_atu, long long* t_tri);

// boxADMM::solve_impl (box_admm.hpp:88-205). Result in w.x (n) and w.y (m+n). h/Alb/Aub/xlb/xub may live in LDS or HBM.
// H(i,j) = H[j*ldh + i], A(r,j) = A[j*lda + r]  (ldh = n, lda = m for plain column-major inputs)
// BIG: the KKT factor is the tiled HBM workspace of pmpc_qp_big.hpp (blocked LDL^T, MFMA trailing updates) instead of the packed triangle.
// JV (BIG only): a block-sparse view of A (JViewRT, pmpc_jview.hpp) — the fused SQP kernel's QPs. With it and `condensed` set the linear algebra runs
// in condensed form: the tiles hold S = H + sigma I + rho_box + A' diag(rho) A (n rows instead of n + m) and every solve is
//     t = r1 + A'(rho o r2),  x = S^{-1} t,  nu = rho o (A x - r2)
// with the two products formed from the view (pmpc_qp_big.hpp, big_build_condensed; CPU restatement: PIVOT_CONDENSED).
// residuals_update with A x and A' y formed from the block-sparse view of A (large-instance kernel, condensed mode): the same products in the same
// order as qp_residuals for finite x, y — the caller tests that — without reading the dense A (two passes over m x n doubles per evaluation)
// ident (round 6): H x from the KKT identity of the last solve instead of a pass over H (n^2 doubles from HBM per evaluation — config C: 512 KB):
//     (H + sigma I + rho_box) x~ + A' nu = r1   =>   H x~ = (r1 - A' nu) - (sigma + rho_box) o x~        (alpha = 1: x = x~)
// r1: the first right-hand side of that solve (the caller saved it in w.t1), nu: its multipliers (w.rhs + n, LDS), A' nu: the fma chain of the solve's first
// product (big_cond_solve) started from 0. Restated by the CPU checker (PIVOT_CONDENSED); see pmpc_qp_cond.hpp for the admission measurements.
template <class JV>
__device__ __forceinline__ void qp_residuals_sparse(const QpLds& w, int n, int m, const double* __restrict__ H, int ldh, const double* __restrict__ h,
                                                   const JV& jv, QpResidualState& r, bool ident = false, double sigma = 0.0) {
    const int ln = lane_id();
    double nAx = 0, nz = 0, nx = 0, rp = 0;
    for (int i0 = 0; i0 < m; i0 += WAVE) {
        const int i = i0 + ln;
        const typename JV::Row rw = jv.rowinfo(i < m ? i : 0);
        double bv[JV::NDER];
        jv.row_block(rw, bv);
        const double a = jv.rowdot_ma(rw, bv, w.x);
        if (i < m) { nAx = fmax(nAx, fabs(a)); nz = fmax(nz, fabs(w.z[i])); rp = fmax(rp, fabs(a - w.z[i])); }
    }
    double nHx = 0, nATy = 0, nh = 0, nyb = 0, rq = 0, rd = 0;
    for (int i0 = 0; i0 < n; i0 += WAVE) {
        const int i = i0 + ln;
        const int ic = i < n ? i : 0;
        const typename JV::Col cc = jv.column(ic);
        double bv[JV::NCB > 0 ? JV::NCB : 1];
        jv.col_block(cc, bv);
        double a;
        if (ident) {
            double atnu;
            if constexpr ((int)JV::NG == 0 && (int)JV::NP == 0) atnu = jv.tab ? jv.coldot_fma_tab(cc, bv, w.rhs + n, 0.0) : jv.coldot_fma(cc, bv, w.rhs + n, 0.0);
            else atnu = jv.coldot_fma(cc, bv, w.rhs + n, 0.0);
            a = w.t1[ic] - atnu;
            a -= (sigma + w.rhob[ic]) * w.x[ic];
        } else a = seq_dot_strided<BIG_MEM_BATCH>(H, (size_t)ldh, 1, ic, n, w.x);
        const double b = jv.coldot_ma(cc, bv, w.y);
        if (i < n) {
            nx = fmax(nx, fabs(w.x[i])); nHx = fmax(nHx, fabs(a)); nATy = fmax(nATy, fabs(b));
            nh = fmax(nh, fabs(h[i])); nyb = fmax(nyb, fabs(w.y[m + i]));
            rq = fmax(rq, fabs(w.x[i] - w.q[i]));
            rd = fmax(rd, fabs(((a + h[i]) + b) + w.y[m + i]));
        }
    }
    nAx = wave_max(fmax(nAx, fmax(nz, nx))); nz = nAx; nx = nAx;
    nHx = wave_max(fmax(fmax(nHx, nATy), fmax(nh, nyb))); nATy = nHx; nh = nHx; nyb = nHx;
    rp = wave_max(rp); rq = wave_max(rq); rd = wave_max(rd);
    r.max_Ax_z_norm = fmax(nAx, fmax(nz, nx));
    r.max_Hx_ATy_h_norm = fmax(nHx, fmax(nATy, fmax(nh, nyb)));
    r.res_prim = rp + rq;
    r.res_dual = rd;
}

struct NoJView {};   // tag: the QP has no structure information (the plain QP entry points)
constexpr int BIG_COND_MAX_ROWS = 272;   // condensed mode: n and m up to this (the passes of its sparse products are unrolled; 4 x 16 ceil(n / 16) doubles of LDS hold a row panel)
template <bool BIG = false, class JV = NoJView, bool SLIM = false, int NW = 1>   // SLIM: the two-wavefronts-per-SIMD build of the large-instance mode (see big_solve); NW = 4: a team of four wavefronts on the condensed linear algebra (BigTeam, pmpc_qp_big.hpp) — this function runs on the first one
__device__ __forceinline__ void boxadmm_solve(QpLds& w, int n, int m, const double* __restrict__ H, int ldh, const double* h,
                                     const double* __restrict__ A, int lda, const double* Alb, const double* Aub, const double* xlb,
                                     const double* xub, const double* x0, const double* y0, const pmpc_qp_settings& s,
                                     pmpc_qp_info& info, long long* tm = nullptr, const JV& jv = JV(), bool condensed = false, BigMail<JV>* mail = nullptr) {   // tm (phase profiling): [0] build + factor, [1] residuals, [2] substitutions, [3] build alone, [4] condensed mode: A'(rho r2), [5] condensed mode: the two triangular passes
    const int ln = lane_id();
    const int N = n + m;
    constexpr bool HASJ = BIG && !std::is_same<JV, NoJView>::value;
    const bool cond = HASJ && condensed;
    // Returns whether the conditioning gate tripped (condensed mode only; PMPC_FLAG_ILLCOND, include/polympc_amd.h): max_i S_ii > PMPC_COND_GATE min_k |d_k|.
    auto build_and_factor = [&](long long* tb) __attribute__((always_inline)) -> bool {   // (inlined at both call sites: as one shared out-of-line function its register use also bounded the occupancy of the two-waves-per-SIMD kernels that call it)
        if constexpr (BIG) {
            if constexpr (HASJ) {
                if (cond) {
                    BigTeam<NW> team;
                    if constexpr (NW > 1) {   // post the routine to the team's helpers (big_helper_loop)
                        if (ln == 0) { mail->op = BIG_OP_FACTOR; mail->n = n; mail->m = m; mail->ldh = ldh; mail->K = w.K; mail->H = H; mail->kdiag = w.kdiag; mail->rho = w.rho; mail->rhs = w.rhs; mail->scr = w.big_lds; mail->jv = jv; }
                        __syncthreads();
                    }
                    const double smax = big_build_condensed<JV, NW>(w.K, n, m, H, ldh, w.kdiag, w.rho, jv, team, NW > 1 ? mail->red4 : nullptr); if (tb) *tb = clock64();
                    const double pmin = big_factor<NW>(w.K, n, w.big_lds, team);
                    return __builtin_amdgcn_readfirstlane((int)(smax > PMPC_COND_GATE * pmin)) != 0;
                }
            }
            big_build(w.K, n, m, H, ldh, A, lda, w.kdiag); if (tb) *tb = clock64(); (void)big_factor<1>(w.K, N, w.big_lds, BigTeam<1>());
        }
        return false;
    };
    // condensed form: S is built from the block-sparse view, which skips the structural zeros of J — with a non-finite model derivative the dense KKT
    // form would propagate NaN through 0 * inf where this one does not: such a solve is flagged whatever x and y look like afterwards
    bool jbad = false;
    if constexpr (HASJ) {
        if (cond) {
            double pr = 0.0;
            for (int i = ln; i < jv.NNo * (int)JV::NX * (int)JV::NDER; i += WAVE) { const int r_ = i / (int)JV::NDER; const double v_ = jv.jblk[r_ * (int)JV::JBS + (i - r_ * (int)JV::NDER)]; pr += v_ - v_; }
            if constexpr ((int)JV::NG > 0) { for (int i = ln; i < jv.NNo * (int)JV::NG * (int)JV::NDER; i += WAVE) { const int r_ = i / (int)JV::NDER; const double v_ = jv.gblk[r_ * (int)JV::JBS + (i - r_ * (int)JV::NDER)]; pr += v_ - v_; } }
            jbad = __builtin_amdgcn_ballot_w64(pr != 0.0) != 0;
        }
    }
    const bool pivoted = !BIG && __builtin_amdgcn_readfirstlane(s.linear_solver) == 1 && w.trp != nullptr;   // Eigen::LDLT's pivoting (LDS-resident mode only)
    auto tick = [&]() -> long long { return tm ? clock64() : 0; };
    // x = x_guess; y = y_guess; z = A*x_guess; q = x_guess  (:97-100)
    for (int i = ln; i < n; i += WAVE) { const double v = x0 ? x0[i] : 0.0; w.x[i] = v; w.q[i] = v; }
    for (int i = ln; i < N; i += WAVE) w.y[i] = y0 ? y0[i] : 0.0;
    wsync();
    for (int i = ln; i < m; i += WAVE) {
        double a = 0.0;
        if (x0) for (int j = 0; j < n; ++j) a += A[(size_t)j * lda + i] * w.x[j];
        w.z[i] = a;
    }
    double rho = s.rho;
    int rho_updates = 1;
    rho_vec_update(w, n, m, Alb, Aub, xlb, xub, rho);
    wsync();
    // K diagonal: (H_ii + sigma) + rho_box ; -1/rho   (:214-222)
    for (int i = ln; i < n; i += WAVE) { double dgl = H[(size_t)i * ldh + i]; dgl += s.sigma; dgl += w.rhob[i]; w.kdiag[i] = dgl; }
    for (int i = ln; i < m; i += WAVE) w.kdiag[n + i] = -w.rhoinv[i];
    wsync();
    bool gave_up = false;   // condensed mode: the conditioning gate tripped — the SQP kernel ends the instance with PMPC_SQP_REDO and the launcher's redo launch solves it in the (n + m)-row form
    { const long long t0 = tick();
      long long t1 = t0;
      if constexpr (BIG) gave_up = build_and_factor(tm ? &t1 : nullptr);
      else { kkt_build(w, n, m, H, ldh, A, lda); t1 = tick(); if (pivoted) kkt_factor_pivoted(w, N); else kkt_factor(w, N); }
      if (tm) { tm[0] += tick() - t0; tm[3] += t1 - t0; } }

    int status = PMPC_QP_UNSOLVED;
    const double alpha = s.alpha;
    QpResidualState rs{0, 0, 1, 1};
    auto residuals = [&]() {
        if constexpr (HASJ) {
            if (cond) {   // (a non-finite iterate takes the dense loops: 0 * inf = NaN on the structural zeros of A)
                double pr = 0.0;
                for (int i = ln; i < n; i += WAVE) pr += w.x[i] - w.x[i];
                for (int i = ln; i < m; i += WAVE) pr += w.y[i] - w.y[i];
                if (__builtin_amdgcn_ballot_w64(pr != 0.0) == 0) { qp_residuals_sparse(w, n, m, H, ldh, h, jv, rs, __builtin_amdgcn_readfirstlane((int)(alpha == 1.0)) != 0, s.sigma); return; }
            }
        }
        qp_residuals(w, n, m, H, ldh, h, A, lda, rs);
    };
    double rho_estimate = 0.0;
    int iter;
    for (iter = 1; iter <= s.max_iter && !gave_up; ++iter) {
        // compute_kkt_rhs (:351-355), z_prev = z
        for (int i = ln; i < n; i += WAVE) w.rhs[i] = ((s.sigma * w.x[i] - h[i]) + w.rhob[i] * w.q[i]) - w.y[m + i];
        if constexpr (HASJ) {   // condensed mode: the residual evaluation behind this iteration (if any) takes H x from the KKT identity and needs r1 (qp_residuals_sparse)
            if (cond && ((s.check_termination != 0 && iter % s.check_termination == 0) || (s.adaptive_rho && iter % s.adaptive_rho_interval == 0)))
                for (int i = ln; i < n; i += WAVE) w.t1[i] = w.rhs[i];
        }
        for (int i = ln; i < m; i += WAVE) { w.zprev[i] = w.z[i]; w.rhs[n + i] = w.z[i] - w.rhoinv[i] * w.y[i]; }
        wsync();
        { const long long t0 = tick();
          if constexpr (BIG) {
              bool done = false;
              if constexpr (HASJ) {
                  if (cond) {   // t = r1 + A'(rho o r2) ; x = S^{-1} t ; nu = rho o (A x - r2)   (big_cond_solve, pmpc_qp_big.hpp)
                      BigTeam<NW> team;
                      if constexpr (NW > 1) { if (ln == 0) mail->op = BIG_OP_SOLVE; __syncthreads(); }
                      big_cond_solve<SLIM, NW, JV>(w.K, n, m, w.rhs, w.rho, w.big_lds, jv, team, tm ? &tm[4] : nullptr, tm ? &tm[5] : nullptr);
                      done = true;
                  }
              }
              if (!done) big_solve<SLIM, 1>(w.K, N, w.rhs, w.big_lds + 256, BigTeam<1>());
          } else { if (pivoted) kkt_solve_pivoted(w, N, w.rhs); else kkt_solve(w, N, w.rhs); }
          if (tm) tm[2] += tick() - t0; }
        for (int i = ln; i < m; i += WAVE) {
            const double zt = w.zprev[i] + w.rhoinv[i] * (w.rhs[n + i] - w.y[i]);
            double zz = alpha * zt;
            zz += (1 - alpha) * w.zprev[i] + w.rhoinv[i] * w.y[i];
            zz = fmin(fmax(zz, Alb[i]), Aub[i]);
            w.z[i] = zz;
            w.y[i] += w.rho[i] * ((alpha * zt + (1 - alpha) * w.zprev[i]) - zz);
        }
        for (int i = ln; i < n; i += WAVE) {
            double xx = alpha * w.rhs[i];
            xx += (1 - alpha) * xx;  // quirk Q1
            w.x[i] = xx;
            double qq = xx + w.rhobinv[i] * w.y[m + i];
            qq = fmin(fmax(qq, xlb[i]), xub[i]);
            w.q[i] = qq;
            w.y[m + i] += w.rhob[i] * (xx - qq);
        }
        wsync();
        const bool check = (s.check_termination != 0 && iter % s.check_termination == 0);
        if (check) {
            const long long t0 = tick();
            residuals();
            if (tm) tm[1] += tick() - t0;
            const double ep = s.eps_abs + s.eps_rel * rs.max_Ax_z_norm, ed = s.eps_abs + s.eps_rel * rs.max_Hx_ATy_h_norm;
            if (rs.res_prim <= ep && rs.res_dual <= ed) { status = PMPC_QP_SOLVED; break; }
        }
        if (s.adaptive_rho && iter % s.adaptive_rho_interval == 0) {
            if (!check) residuals();
            const double rpn = rs.res_prim / (rs.max_Ax_z_norm + DIV_BY_ZERO_REGUL);
            const double rdn = rs.res_dual / (rs.max_Hx_ATy_h_norm + DIV_BY_ZERO_REGUL);
            double new_rho = rho * sqrt(rpn / (rdn + DIV_BY_ZERO_REGUL));
            new_rho = fmax(RHO_MIN, fmin(new_rho, RHO_MAX));
            rho_estimate = new_rho;
            if (new_rho < rho / s.adaptive_rho_tolerance || new_rho > rho * s.adaptive_rho_tolerance) {
                // rho_vec_update + update_kkt_rho (:448-452) + refactor
                for (int i = ln; i < n; i += WAVE) w.t2[i] = w.rhob[i];
                wsync();
                rho = new_rho;
                rho_vec_update(w, n, m, Alb, Aub, xlb, xub, rho);
                ++rho_updates;
                wsync();
                for (int i = ln; i < n; i += WAVE) w.kdiag[i] += (w.rhob[i] - w.t2[i]);
                for (int i = ln; i < m; i += WAVE) w.kdiag[n + i] = -w.rhoinv[i];
                wsync();
                if constexpr (BIG) { gave_up = build_and_factor(nullptr); if (gave_up) break; }
                else { kkt_build(w, n, m, H, ldh, A, lda); if (pivoted) kkt_factor_pivoted(w, N); else kkt_factor(w, N); }
            }
        }
    }
    if (iter > s.max_iter) status = PMPC_QP_MAX_ITER_EXCEEDED;
    double worst = 0.0;   // NaN-propagating: |x| summed through max() would drop NaNs, (x - x) is 0 for finite x and NaN otherwise
    for (int i = ln; i < n; i += WAVE) worst += fabs(w.x[i] - w.x[i]);
    for (int i = ln; i < N; i += WAVE) worst += fabs(w.y[i] - w.y[i]);
    const bool bad = __builtin_amdgcn_ballot_w64(worst != 0.0) != 0;
    info.status = status; info.iter = iter; info.rho_updates = rho_updates; info.flags = ((bad || jbad) ? PMPC_FLAG_NONFINITE : 0) | (gave_up ? PMPC_FLAG_ILLCOND : 0);
    info.rho_estimate = rho_estimate; info.res_prim = rs.res_prim; info.res_dual = rs.res_dual;
}

}  // namespace pmpc
