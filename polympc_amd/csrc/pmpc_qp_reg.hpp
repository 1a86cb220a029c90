// polympc_amd — register-resident box-ADMM QP solve for compile-time sizes with n+m <= 64 (one wavefront per QP).
//
// Same algorithm, constants and update order as pmpc_qp.hpp (boxADMM::solve_impl, box_admm.hpp:88-205); the linear solve
// of every ADMM iteration is organised differently (a different, equally static, order of floating-point operations —
// the test suite checks it bit for bit against a CPU restatement of exactly this order):
//   * the KKT matrix is INVERTED once per factorisation point (first iteration and every accepted rho update) with the
//     symmetric sweep operator in MFMA accumulator tiles (RegKkt::invert), and lane i keeps row i of W = -K^{-1} in a
//     register array a[N] (compile-time register indices after full unrolling);
//   * the per-iteration solve is then the mat-vec  x_i = -sum_j a[j] * rhs_j : the rhs entries are broadcast with
//     v_readlane and all 56 products are independent — no 2N-step substitution chain in the ADMM loop;
//   * all ADMM vectors are one register per lane: lanes [0,n) carry x, q, y_box, rho_box, h, xlb, xub; lanes
//     [n,n+m) carry z, y_a, rho, Alb, Aub. The ADMM iteration therefore runs entirely out of registers.
//   * H and A are read from HBM/L2 (coalesced down columns) only to build K and, every check_termination-th
//     iteration, for the residual mat-vecs.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_qp.hpp"

namespace pmpc {

__device__ __forceinline__ double bcast_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// c <- fma(-a, x, c) on the lanes ABOVE `j` only (x wave-uniform, in SGPRs). The lane mask is a compile-time constant, so
// it is applied with one scalar shift into EXEC instead of a compare + two selects; EXEC is restored inside the statement
// (the compiler never sees a modified EXEC; SCC, which the scalar shift overwrites, is declared clobbered). s_nop 0 completes the v_readlane(SGPR write) -> VALU(SGPR read) wait states.
__device__ __forceinline__ double fnma_lanes_above(double c, double a, double x_uniform, int j) {
    asm("s_lshl_b64 exec, -1, %3\n\ts_nop 0\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_mov_b64 exec, -1" : "+v"(c) : "v"(a), "s"(x_uniform), "i"(j + 1) : "scc");
    return c;
}
__device__ __forceinline__ double fnma_lanes_below(double c, double a, double x_uniform, int j) {
    asm("s_lshr_b64 exec, -1, %3\n\ts_nop 0\n\tv_fma_f64 %0, -%1, %2, %0\n\ts_mov_b64 exec, -1" : "+v"(c) : "v"(a), "s"(x_uniform), "i"(64 - j) : "scc");
    return c;
}
// dst <- src on the lanes BELOW `j` only
__device__ __forceinline__ double mov_lanes_below(double dst, double src, int j) {
    asm("s_lshr_b64 exec, -1, %2\n\tv_mov_b64 %0, %1\n\ts_mov_b64 exec, -1" : "+v"(dst) : "v"(src), "i"(64 - j) : "scc");
    return dst;
}

// One wavefront per workgroup: DS operations of a wave are executed in issue order, so a write followed by a read of the
// same LDS address needs no s_waitcnt / s_barrier — only the compiler must not reorder them.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }
// scheduling fence: keeps the machine scheduler from hoisting dozens of v_readlane broadcasts (SGPR pairs) or LDS loads
// across phase boundaries, which otherwise inflates the register demand far beyond the algorithm's live set
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// dst <- src on lanes [lo, hi) only (compile-time range, hi > lo)
__device__ __forceinline__ double mov_lanes_range(double dst, double src, int lo, int hi) {
    asm("s_bfm_b64 exec, %2, %3\n\tv_mov_b64 %0, %1\n\ts_mov_b64 exec, -1" : "+v"(dst) : "v"(src), "i"(hi - lo), "i"(lo));
    return dst;
}

// a0..a6 <- 0 on lane `k` only (compile-time k): one EXEC switch for the seven moves
__device__ __forceinline__ void zero_on_lane(double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, int k) {
    asm("s_lshl_b64 exec, 1, %7\n\tv_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, 0\n\tv_mov_b64 %4, 0\n\tv_mov_b64 %5, 0\n\t"
        "v_mov_b64 %6, 0\n\ts_mov_b64 exec, -1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : "i"(k) : "scc");
}

template <int N>
struct RegKkt {
    double a[N];  // row `lane` of W = -K^{-1}

    using d4 = double __attribute__((ext_vector_type(4)));
    static constexpr int BK = 8;                      // pivots swept per block
    static constexpr int NB = (N + BK - 1) / BK;      // number of blocks
    static constexpr int NT = (N + 15) / 16;          // 16x16 tiles per dimension
    static constexpr int NP = NT * 16;                // padded dimension
    // LDS staging, two layouts chosen so that every access pattern below is bank-conflict free:
    //  * k-major operand panels PA and PB: element (t, row) at t*SK + row. Written row-per-lane (consecutive lanes ->
    //    consecutive doubles), read in the MFMA operand pattern (lane>>4)*SK + (lane&15): SK = 16 (mod 32) puts the
    //    32 lanes of a half-wave on 32 distinct 8-byte banks.
    //  * row-major exchange buffer X: element (row, t) at row*SX + t with SX = 9. Accumulator-tile side: (lane>>4)*SX +
    //    (lane&7) (+ const), row-per-lane side: lane*SX + t — both spread over the banks (an odd stride).
    //    X aliases PB: a wave's DS operations execute in issue order, and X is never live at the same time as PB.
    //  Both are sized for all 64 lanes (idle lanes >= N store too; their values are never consumed by live rows).
    static constexpr int SK = 80;
    static constexpr int SX = BK + 1;
    static constexpr int XSZ = (64 * SX > BK * SK) ? 64 * SX : BK * SK;
    static constexpr int TRI = BK * SK + XSZ;         // doubles of LDS staging
    static_assert(N <= 64 && SK % 32 == 16 && SK >= 64, "panel stride");

    // W = -K^{-1} by the symmetric sweep operator, static pivot order (K is quasi-definite: every pivot is non-zero),
    // in blocks of BK = 8 pivots. The matrix lives in 16x16 fp64 MFMA accumulator tiles T[R][C], C <= R (block-lower
    // storage: a tile above the block diagonal is the transpose of its mirror image and is never materialised).
    // Block step on pivots kb..kb+7:
    //   1. the panel M[:, block] goes tiles -> exchange buffer -> row-per-lane registers p[8] (rows above the pivot tile
    //      row come out of the pivot tile ROW, transposed)
    //   2. PB <- old panel (B operand)
    //   3. in-panel scalar sweeps (v_readlane broadcasts of the pivot row):  r = 1/p_k[t];  l_i = p_i[t]*r (i != k), l_k = -r;
    //        u != t:  p_k[u] <- 0, then p_i[u] <- fma(-l_i, pivotrow[u], p_i[u]) on every lane (lane k: = pivotrow[u]*r);  p[t] <- l
    //   4. PA <- -p (A operand); rows of the block are zero in PA and PB, so the update leaves block rows / columns alone
    //   5. stored tiles:  T[R][C] <- T[R][C] + PA_R * PB_C^T  (two v_mfma_f64_16x16x4_f64 each; the instruction is a
    //      k-ascending fma chain — verified on gfx950, tests/experiments/mfma_f64_probe.hip — so every entry receives
    //      fma(-p_i[t], old_j[t], m_ij) for t ascending, which is what the CPU checker of the test suite restates)
    //   6. write-back: M[:, block] <- p into the pivot tile column, then M[block, :] <- p^T into the pivot tile row
    // Finally the tiles are converted to row-per-lane registers a[] for the mat-vec.
    // kcol(j, z) returns K(lane, j) for j != lane, needed for j <= 16*(lane/16)+15 only (z: see below); it is called 8
    // columns at a time, one group ahead of use. diag = K(lane, lane).
    template <class KCol>
    __device__ __forceinline__ void invert(int ln_in, double* st, double diag, KCol kcol) {
        int ln = ln_in;
        asm volatile("" : "+v"(ln));   // keep the lane predicates below local to this function (no hoisting into long-lived SGPR masks)
        double* PA = st;
        double* PB = st + BK * SK;
        double* X = PB;
        const int lr = ln >> 4, lc = ln & 15;
        d4 T[NT][NT];   // only C <= R is used
        // row layout -> accumulator tiles, 8 columns at a time (loads of the next group are in flight while this one is staged)
        // `z` is an opaque zero that kcol adds to its addresses: redefining it once per group pins each group's loads
        // behind the previous group's staging (loads from read-only kernel arguments may otherwise be hoisted to the
        // top, all 2N registers at once)
        double cur[BK], nxt[BK];
        int z = 0;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int t = 0; t < BK; ++t) cur[t] = (t < N) ? kcol(t, z) : 0.0;
#pragma unroll
        for (int g = 0; g < NP / BK; ++g) {
            asm volatile("" : "+v"(z) :: "memory");
#pragma unroll
            for (int t = 0; t < BK; ++t) nxt[t] = ((g + 1) * BK + t < N) ? kcol(((g + 1) * BK + t < N) ? (g + 1) * BK + t : 0, z) : 0.0;
#pragma unroll
            for (int t = 0; t < BK; ++t) X[ln * SX + t] = cur[t];
            lds_order();
            if ((ln >> 3) == g) X[ln * SX + (ln & 7)] = diag;      // the diagonal entries of this column group
            lds_order();
            if ((lc >> 3) == (g % 2)) {
#pragma unroll
                for (int R = g / 2; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[R][g / 2][r] = X[(16 * R + lr + 4 * r) * SX + (lc & 7)];
            }
            lds_order();
            sched_fence();
#pragma unroll
            for (int t = 0; t < BK; ++t) cur[t] = nxt[t];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int kb = b * BK;
            const int Cb = kb / 16, hb = (kb % 16) / BK;
            // 1. panel -> row-per-lane registers: rows of tile rows >= Cb from tile column Cb (tile-local columns
            //    [8*hb, 8*hb+8)), rows of tile rows < Cb from tile row Cb (tile-local rows [8*hb, 8*hb+8), transposed)
            if ((lc >> 3) == hb) {
#pragma unroll
                for (int R = Cb; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(16 * R + lr + 4 * r) * SX + (lc & 7)] = T[R][Cb][r];
            }
#pragma unroll
            for (int C = 0; C < Cb; ++C)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) X[(16 * C + lc) * SX + lr + 4 * rr] = T[Cb][C][2 * hb + rr];
            lds_order();
            double p[BK];
#pragma unroll
            for (int t = 0; t < BK; ++t) p[t] = X[ln * SX + t];
            lds_order();
            const bool inb = (ln >> 3) == b;
            // 2. B operand: the panel as it was at the start of the block
#pragma unroll
            for (int t = 0; t < BK; ++t) PB[t * SK + ln] = (inb || kb + t >= N) ? 0.0 : p[t];
            // 3. in-panel sweeps
#pragma unroll
            for (int t = 0; t < BK; ++t) {
                const int k = kb + t;
                if (k < N) {
                    const double dk = bcast_lane(p[t], k);
                    const double r = 1.0 / dk;
                    double rk[BK];
#pragma unroll
                    for (int u = 0; u < BK; ++u) rk[u] = (u != t) ? bcast_lane(p[u], k) : 0.0;
                    const double l = (ln == k) ? -r : p[t] * r;
                    // the seven other columns of lane k start from zero, so that one fma serves every lane
                    zero_on_lane(p[(t + 1) & 7], p[(t + 2) & 7], p[(t + 3) & 7], p[(t + 4) & 7], p[(t + 5) & 7], p[(t + 6) & 7], p[(t + 7) & 7], k);
#pragma unroll
                    for (int u = 0; u < BK; ++u)
                        if (u != t) p[u] = fma(-l, rk[u], p[u]);
                    p[t] = l;
                    sched_fence();
                }
            }
            // 4. A operand
#pragma unroll
            for (int t = 0; t < BK; ++t) PA[t * SK + ln] = (inb || kb + t >= N) ? 0.0 : -p[t];
            lds_order();
            // 5. rank-8 update of every stored tile
#pragma unroll
            for (int s2 = 0; s2 < BK / 4; ++s2) {
                double av[NT], bv[NT];
#pragma unroll
                for (int R = 0; R < NT; ++R) {
                    av[R] = PA[(4 * s2 + lr) * SK + 16 * R + lc];
                    bv[R] = PB[(4 * s2 + lr) * SK + 16 * R + lc];
                }
#pragma unroll
                for (int R = 0; R < NT; ++R)
#pragma unroll
                    for (int C = 0; C <= R; ++C) T[R][C] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[R], bv[C], T[R][C], 0, 0, 0);
                sched_fence();
            }
            lds_order();
            // 6. write-back of the swept panel: pivot tile column, then pivot tile row (the diagonal block ends up as the transpose of p)
#pragma unroll
            for (int t = 0; t < BK; ++t) X[ln * SX + t] = p[t];
            lds_order();
            if ((lc >> 3) == hb) {
#pragma unroll
                for (int R = Cb; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[R][Cb][r] = X[(16 * R + lr + 4 * r) * SX + (lc & 7)];
            }
#pragma unroll
            for (int C = 0; C <= Cb; ++C)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) T[Cb][C][2 * hb + rr] = X[(16 * C + lc) * SX + lr + 4 * rr];
            lds_order();
            sched_fence();
        }
        // accumulator tiles -> row-per-lane registers (columns of tile rows above the diagonal come from the mirror tiles)
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            if ((lc >> 3) == (g % 2)) {
#pragma unroll
                for (int R = g / 2; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(16 * R + lr + 4 * r) * SX + (lc & 7)] = T[R][g / 2][r];
            }
#pragma unroll
            for (int C = 0; C < g / 2; ++C)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) X[(16 * C + lc) * SX + lr + 4 * rr] = T[g / 2][C][2 * (g % 2) + rr];
            lds_order();
#pragma unroll
            for (int t = 0; t < BK; ++t)
                if (g * BK + t < N) a[g * BK + t] = X[ln * SX + t];
            lds_order();
        }
    }

    // K^{-1} c, one entry per lane:  -(W c) with four interleaved partial sums (j mod 4), combined as (s0+s1)+(s2+s3)
    __device__ __forceinline__ double apply(double c) const {
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < N; j0 += 8) {
            double t[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) t[jj] = (j0 + jj < N) ? bcast_lane(c, (j0 + jj < N) ? j0 + jj : 0) : 0.0;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                if (j < N) {
                    if ((j & 3) == 0) acc0 = fma(a[j], t[jj], acc0);
                    if ((j & 3) == 1) acc1 = fma(a[j], t[jj], acc1);
                    if ((j & 3) == 2) acc2 = fma(a[j], t[jj], acc2);
                    if ((j & 3) == 3) acc3 = fma(a[j], t[jj], acc3);
                }
            }
            // groups of 8 broadcasts: the empty statement ties the next group's v_readlane to this group's results, which
            // keeps instruction selection from hoisting all 2N scalar broadcasts (more SGPRs than exist) to the top
            asm volatile("" : "+v"(c), "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));
        }
        return -((acc0 + acc1) + (acc2 + acc3));
    }
};

// boxADMM::solve_impl for compile-time (NN, MM); h/Alb/Aub/xlb/xub/x0/y0: LDS or HBM pointers; result -> out_x (NN), out_y (MM+NN)
// tr: LDS staging of RegKkt<NN+MM>::TRI doubles.
// STACKED: H and A are the upper / lower block of ONE (n+m) x n column-major array (leading dimension n+m, A = H + n):
// every lane then reads row `lane` of that array with a compile-time stride, i.e. one base address + immediate offsets.
template <int NN, int MM, bool STACKED = false>
__device__ __forceinline__ void boxadmm_solve_reg(const double* __restrict__ H, const double* h, const double* __restrict__ A,
                                                  const double* Alb, const double* Aub, const double* xlb, const double* xub,
                                                  const double* x0, const double* y0, const pmpc_qp_settings& s, pmpc_qp_info& info,
                                                  double* out_x, double* out_y, double* tr, long long* dbg = nullptr) {
    constexpr int N = NN + MM;
    static_assert(N <= WAVE, "register-resident path needs n+m <= 64");
    const int ln = lane_id();
    const bool isP = ln < NN;
    const bool isC = (ln >= NN) && (ln < N);
    const int r = isC ? ln - NN : 0;

    // per-lane problem data
    // NOTE: lane-predicated code in this function is written branch-free (clamped unconditional loads + selects):
    // hipcc (ROCm 7.2) may place VGPR spills inside the partial-EXEC "Flow" blocks of a divergent if/else, which loses the
    // inactive lanes' copies of values that are live across the branch.
    const int lp = isP ? ln : 0;                 // clamped primal index
    const double hv = isP ? h[lp] : 0.0;
    const double lo = isP ? xlb[lp] : (isC ? Alb[r] : 0.0);
    const double hi = isP ? xub[lp] : (isC ? Aub[r] : 0.0);
    const int type = classify_bounds(lo, hi);

    // Row `lane` of [H ; A] and column `lane` of A are read with unconditional, clamped addresses. `zo` is an opaque zero
    // (redefined by an empty asm statement next to each use): without it the address arithmetic of all 2N loads — and,
    // for read-only kernel arguments, the loads themselves — is hoisted out of the ADMM loops and spilled.
    //   STACKED: one uniform base (H) + 32-bit per-lane element offsets; otherwise a per-lane base and stride.
    const double* rowp = isP ? (H + ln) : (A + r);
    const int rstride = isP ? NN : (isC ? MM : 0);
    const double* colA = A + (size_t)lp * MM;
    const unsigned roff = (ln < N) ? ln : 0, coff = lp * N + NN;
    auto Krow = [&](int j, int zo) -> double {   // (H or A)(row of this lane, j), j < NN
        if constexpr (STACKED) return H[(unsigned)(roff + zo) + (unsigned)(j * N)];
        else return rowp[(size_t)j * (size_t)(unsigned)(rstride + zo)];
    };
    auto Acol = [&](int k, int zo) -> double {   // A(k, lane), k < MM (primal lanes)
        if constexpr (STACKED) return H[(unsigned)(coff + zo) + (unsigned)k];
        else return colA[k + zo];
    };
    constexpr int LDH = STACKED ? N : NN;

    // state: xv = x (primal lanes) / z (constraint lanes); yv = y_box / y_a; qv = q (primal lanes)
    double xv = 0.0, yv = 0.0, qv = 0.0;
    {
        const double x0v = x0 ? x0[lp] : 0.0, ybv = y0 ? y0[MM + lp] : 0.0, yav = y0 ? y0[r] : 0.0;
        xv = isP ? x0v : 0.0; qv = xv; yv = isP ? ybv : (isC ? yav : 0.0);
    }
    if (x0) {  // z = A * x_guess
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NN; ++j) acc += Krow(j, 0) * bcast_lane(xv, j);
        xv = isC ? acc : xv;
    }

    double rho = s.rho;
    int rho_updates = 1;
    double rhov = rho_of(type, rho);
    double rhoinv = 1.0 / rhov;
    double kdiag;
    {
        double kd = H[(size_t)lp * LDH + lp]; kd += s.sigma; kd += rhov;
        kdiag = isP ? kd : -rhoinv;
    }

    RegKkt<N> K;
    int status = PMPC_QP_UNSOLVED;
    const double alpha = s.alpha;
    double max_Ax_z_norm = 0.0, max_Hx_ATy_h_norm = 0.0, res_prim = 1.0, res_dual = 1.0, rho_estimate = 0.0;
    // Outer loop = one KKT factorisation (first pass and after every accepted rho update); inner loop = ADMM iterations
    // on that factor. Keeping the high-register-pressure factorisation OUT of the inner loop keeps the inner loop's
    // state in registers (no scratch traffic inside the ADMM iterations).
    int iter = 1;
    int until_check = s.check_termination, until_adapt = s.adaptive_rho_interval;   // iterations left until the next multiple
    bool running = true;
    while (running) {
        {   // construct_kkt_matrix + factorise_kkt_matrix
            const long long f0 = dbg ? clock64() : 0;
            K.invert(ln, tr, kdiag, [&](int j, int z) -> double {   // row `lane` of [H  A^T ; A  .] (construct_kkt_matrix, box_admm.hpp:209-223)
                if (j < NN) return Krow(j < NN ? j : 0, z);
                const double v = Acol(j >= NN ? j - NN : 0, z);
                return isP ? v : 0.0;
            });
            if (dbg) dbg[0] += clock64() - f0;
        }
        bool refactor = false;
        for (; iter <= s.max_iter; ++iter) {
            const double zprev = xv;  // meaningful on constraint lanes
            const double rhsP = ((s.sigma * xv - hv) + rhov * qv) - yv;
            const double rhsC = xv - rhoinv * yv;
            const double rhs = isP ? rhsP : (isC ? rhsC : 0.0);
            const double sol = K.apply(rhs);
            // both role updates are evaluated on every lane and selected (branch-free)
            const double zt = zprev + rhoinv * (sol - yv);
            double zz = alpha * zt;
            zz += (1 - alpha) * zprev + rhoinv * yv;
            zz = fmin(fmax(zz, lo), hi);
            const double yC = yv + rhov * ((alpha * zt + (1 - alpha) * zprev) - zz);
            double xx = alpha * sol;
            xx += (1 - alpha) * xx;  // quirk Q1
            double qq = xx + rhoinv * yv;
            qq = fmin(fmax(qq, lo), hi);
            const double yP = yv + rhov * (xx - qq);
            xv = isP ? xx : (isC ? zz : xv);
            qv = isP ? qq : qv;
            yv = isP ? yP : (isC ? yC : yv);
            // iter % check_termination == 0 / iter % adaptive_rho_interval == 0 (box_admm.hpp:141,:160) as countdowns
            bool check = false, adapt = false;
            if (s.check_termination != 0 && --until_check == 0) { check = true; until_check = s.check_termination; }
            if (s.adaptive_rho && --until_adapt == 0) { adapt = true; until_adapt = s.adaptive_rho_interval; }
            if (check || adapt) {  // residuals_update, box_admm.hpp:398-415
                const long long r0 = dbg ? clock64() : 0;
                // loads in chunks of RC columns (independent, coalesced), each followed by its slice of the mat-vec chain
                constexpr int RC = 12;
                int zr = 0;            // opaque zero added to the addresses: keeps these loop-invariant loads inside the loop
                asm volatile("" : "+v"(zr));
                double acc = 0.0;      // lanes < n: (H x)_i ; lanes in [n, N): (A x)_r
#pragma unroll
                for (int j0 = 0; j0 < NN; j0 += RC) {
                    double mrow[RC];
#pragma unroll
                    for (int j = 0; j < RC; ++j) mrow[j] = (j0 + j < NN) ? Krow(j0 + j < NN ? j0 + j : 0, zr) : 0.0;
#pragma unroll
                    for (int j = 0; j < RC; ++j) if (j0 + j < NN) acc += mrow[j] * bcast_lane(xv, j0 + j);
                    sched_fence();
                }
                double aty = 0.0;      // lanes < n: (A^T y_a)_i
#pragma unroll
                for (int k0 = 0; k0 < MM; k0 += RC) {
                    double mcol[RC];
#pragma unroll
                    for (int k = 0; k < RC; ++k) mcol[k] = (k0 + k < MM) ? Acol((k0 + k < MM) ? k0 + k : 0, zr) : 0.0;
#pragma unroll
                    for (int k = 0; k < RC; ++k) if (k0 + k < MM) aty += mcol[k] * bcast_lane(yv, NN + k0 + k);
                    sched_fence();
                }
                const double nAx = wave_max(isC ? fabs(acc) : 0.0), nz = wave_max(isC ? fabs(xv) : 0.0), nx = wave_max(isP ? fabs(xv) : 0.0);
                const double rp = wave_max(isC ? fabs(acc - xv) : 0.0), rq = wave_max(isP ? fabs(xv - qv) : 0.0);
                const double nHx = wave_max(isP ? fabs(acc) : 0.0), nATy = wave_max(isP ? fabs(aty) : 0.0);
                const double nh = wave_max(fabs(hv)), nyb = wave_max(isP ? fabs(yv) : 0.0);
                const double rd = wave_max(isP ? fabs(((acc + hv) + aty) + yv) : 0.0);
                max_Ax_z_norm = fmax(nAx, fmax(nz, nx));
                max_Hx_ATy_h_norm = fmax(nHx, fmax(nATy, fmax(nh, nyb)));
                res_prim = rp + rq;
                res_dual = rd;
                if (dbg) dbg[1] += clock64() - r0;
            }
            if (check) {
                const double ep = s.eps_abs + s.eps_rel * max_Ax_z_norm, ed = s.eps_abs + s.eps_rel * max_Hx_ATy_h_norm;
                if (__builtin_amdgcn_readfirstlane((int)(res_prim <= ep && res_dual <= ed))) { status = PMPC_QP_SOLVED; running = false; break; }
            }
            if (adapt) {
                const double rpn = res_prim / (max_Ax_z_norm + DIV_BY_ZERO_REGUL);
                const double rdn = res_dual / (max_Hx_ATy_h_norm + DIV_BY_ZERO_REGUL);
                double new_rho = rho * ::sqrt(rpn / (rdn + DIV_BY_ZERO_REGUL));
                new_rho = fmax(RHO_MIN, fmin(new_rho, RHO_MAX));
                rho_estimate = new_rho;
                if (__builtin_amdgcn_readfirstlane((int)(new_rho < rho / s.adaptive_rho_tolerance || new_rho > rho * s.adaptive_rho_tolerance))) {
                    const double prev = rhov;
                    rho = new_rho;
                    rhov = rho_of(type, rho);
                    rhoinv = 1.0 / rhov;
                    ++rho_updates;
                    kdiag = isP ? (kdiag + (rhov - prev)) : -rhoinv;   // update_kkt_rho, box_admm.hpp:448-452
                    refactor = true;
                    ++iter;
                    break;
                }
            }
        }
        if (!refactor) running = false;
    }
    if (iter > s.max_iter) status = PMPC_QP_MAX_ITER_EXCEEDED;
    if (isP) { out_x[ln] = xv; out_y[MM + ln] = yv; }
    if (isC) out_y[r] = yv;
    info.status = status; info.iter = iter; info.rho_updates = rho_updates; info._pad = 0;
    info.rho_estimate = rho_estimate; info.res_prim = res_prim; info.res_dual = res_dual;
}

}  // namespace pmpc
