// polympc_amd — register-resident box-ADMM QP solve for compile-time sizes with n+m <= 64 (one wavefront per QP).
//
// Same algorithm, constants and update order as pmpc_qp.hpp (boxADMM::solve_impl, box_admm.hpp:88-205); the linear solve
// of every ADMM iteration is organised differently (a different, equally static, order of floating-point operations —
// the test suite checks it bit for bit against a CPU restatement of exactly this order):
//   * the KKT matrix is INVERTED once per factorisation point (first iteration and every accepted rho update) with the
//     symmetric sweep operator in MFMA accumulator tiles (RegKkt::invert), and every lane keeps a 4 x 16 slice of
//     W = -K^{-1} in a register array (compile-time register indices after full unrolling);
//   * the per-iteration solve is then the mat-vec  x = -W rhs : lane 16r+c multiplies the 16 columns of block r (rhs entries
//     broadcast inside the 16-lane row by the DPP modifier of v_fmac_f64 — no v_readlane, no SGPRs) against rows c, c+16,
//     c+32, c+48, and two permlane swaps add the four block partial sums of every row — no 2N-step substitution chain
//     in the ADMM loop;
//   * all ADMM vectors are one register per lane: lanes [0,n) carry x, q, y_box, rho_box, h, xlb, xub; lanes
//     [n,n+m) carry z, y_a, rho, Alb, Aub. The ADMM iteration therefore runs entirely out of registers.
//   * H and A are read from HBM/L2 (coalesced down columns) only to build K and, every check_termination-th
//     iteration, for the residual mat-vecs.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_qp.hpp"

#ifndef PMPC_REG_RESIDUAL_BATCH
#define PMPC_REG_RESIDUAL_BATCH 14
#endif

namespace pmpc {

__device__ __forceinline__ double bcast_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// One wavefront per workgroup: DS operations of a wave are executed in issue order, so a write followed by a read of the
// same LDS address needs no s_waitcnt / s_barrier — only the compiler must not reorder them.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }
// scheduling fence: keeps the machine scheduler from hoisting dozens of v_readlane broadcasts (SGPR pairs) or LDS loads
// across phase boundaries, which otherwise inflates the register demand far beyond the algorithm's live set
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// a0..a6 <- 0 and l <- lk on lane `k` only (compile-time k): one EXEC switch for the eight moves.
// The statement narrows EXEC to one lane and restores it to the CONSTANT all-ones, with EXEC not in the clobber list: sound exactly where the statement is
// emitted with every lane enabled. That is a property of the BUILT code, and it is checked there: tests/tools_exec_regions.py follows EXEC (and every saved
// copy of it, through SGPR spill lanes too) over the control-flow graph of each shipped kernel and proves that every one of these bodies starts at full EXEC
// (tests/test_kernel_occupancy_cpu.py). Round 6 examined it as the suspect behind round 5's miscompiled hook build (EXPERIMENTS.md): refuted — in that build
// too every body starts at full EXEC, and the fault is identical with each of the forms below (it is a VGPR -> AGPR live-range split of the lane id that the
// compiler placed inside the else-block of a divergent if / else: the same test now checks the built code for that pattern as well).
// PMPC_PIVOT_SETUP_FORM (developer switch of that experiment): 0 the shipped form; 1 the same as `asm volatile`; 2 EXEC saved in an SGPR pair and restored
// from it (correct under any EXEC); 3 no EXEC write at all: v_cndmask selects on lane == k.
#ifndef PMPC_PIVOT_SETUP_FORM
#define PMPC_PIVOT_SETUP_FORM 0
#endif
__device__ __forceinline__ void pivot_lane_setup(double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double& l,
                                                 double lk, int k) {
    asm("s_lshl_b64 exec, 1, %9\n\tv_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, 0\n\tv_mov_b64 %4, 0\n\tv_mov_b64 %5, 0\n\t"
        "v_mov_b64 %6, 0\n\tv_mov_b64 %7, %8\n\ts_mov_b64 exec, -1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(l) : "v"(lk), "i"(k) : "scc");
}
__device__ __forceinline__ void pivot_lane_setup(double& a0, double& a1, double& a2, double& l, double lk, int k) {
#if PMPC_PIVOT_SETUP_FORM == 1
    asm volatile("s_lshl_b64 exec, 1, %5\n\tv_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, %4\n\ts_mov_b64 exec, -1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(l) : "v"(lk), "i"(k) : "scc");
#elif PMPC_PIVOT_SETUP_FORM == 2
    unsigned long long saved, win;
    asm volatile("s_mov_b64 %4, exec\n\ts_lshl_b64 %5, 1, %7\n\ts_and_b64 exec, %4, %5\n\tv_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, %6\n\ts_mov_b64 exec, %4"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(l), "=&s"(saved), "=&s"(win) : "v"(lk), "i"(k) : "scc");
#elif PMPC_PIVOT_SETUP_FORM == 3
    const bool me = lane_id() == k;
    a0 = me ? 0.0 : a0; a1 = me ? 0.0 : a1; a2 = me ? 0.0 : a2; l = me ? lk : l;
#else
    asm("s_lshl_b64 exec, 1, %5\n\tv_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, %4\n\ts_mov_b64 exec, -1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(l) : "v"(lk), "i"(k) : "scc");
#endif
}
// 1 / d, branch-free: the generic IEEE division expansion (v_rcp_f64, two Newton steps, residual correction, v_div_fixup for
// 0 / inf / NaN) WITHOUT its v_div_scale pre-scaling — 8 VALU operations instead of 12. The pre-scaling only matters when d or
// 1/d is subnormal (|d| beyond 2^+-1021); everywhere else the result is the correctly rounded quotient, bit for bit what the
// CPU checker's 1.0 / d gives. (A scalar range test + branch to the generic division was measured: the branch latency in
// front of every pivot costs more than the four operations it saves.)
__device__ __forceinline__ double recip_uniform(double d) {
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    return __builtin_amdgcn_div_fixup(y, d, 1.0);
}

// acc <- fma(x of lane J of this lane's 16-lane row, w, acc): the broadcast is a DPP operand modifier of the fma itself
// (row_newbcast, the one DPP control 64-bit VALU operations accept), so a mat-vec needs no v_readlane / SGPR traffic.
template <int J>
__device__ __forceinline__ double fmac_rowbcast(double acc, double x, double w) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w), "i"(J));
    return acc;
}
// a <-> b exchanges: swap32 trades a's lanes 32..63 with b's lanes 0..31, swap16 a's odd 16-lane rows with b's even rows
__device__ __forceinline__ void swap32(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]); b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]); b = __hiloint2double(hi[1], lo[1]);
}

template <int N>
struct RegKkt {
    // W = -K^{-1} in mat-vec layout: lane 16*r + c holds  a[16*q + j] = W(16*q + c, 16*r + j)  (q < NT, j < 16; zero where the
    // column index is >= N): the 16 columns of block r against the rows c, c+16, c+32, ... — see apply()
    double a[((N + 15) / 16) * 16];
    using d4 = double __attribute__((ext_vector_type(4)));
    static constexpr int BK = 4;                      // pivots swept per block (4: one MFMA k-step; the scalar in-panel sweep costs
                                                      // 16 + 2*BK broadcast/setup operations per pivot next to BK-1 useful fma — measured
                                                      // against BK = 8: 1000 fewer VALU operations per inverse, same number of MFMA)
    static constexpr int SG = 8;                      // columns per group of the initial row -> tile staging
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
    static constexpr int SX = SG + 1;
    static constexpr int XSZ = (64 * SX + 64 > BK * SK) ? 64 * SX + 64 : BK * SK;   // 64 exchange rows + the diagonal slots
    static constexpr int SY = 78;                     // final conversion buffer Y, see below
    static constexpr int TRI = (BK * SK + XSZ > 16 * SY) ? BK * SK + XSZ : 16 * SY;   // doubles of LDS staging
    static_assert(N <= 64 && SK % 32 == 16 && SK >= 64, "panel stride");
    // final conversion buffer Y: 16 rows of one tile row, all columns, row stride SY = 78 (14 mod 32, 2 mod 4): the mirror-tile
    // writes and the row reads are bank-conflict free, the direct-tile writes collide on 2 of 32 banks
    static_assert(16 * SY <= TRI && BK % 4 == 0 && 16 % BK == 0 && BK <= SG, "conversion buffer fits the staging; block size divides a tile");

    // W = -K^{-1} by the symmetric sweep operator, static pivot order (K is quasi-definite: every pivot is non-zero),
    // in blocks of BK pivots. The matrix lives in 16x16 fp64 MFMA accumulator tiles T[R][C], C <= R (block-lower
    // storage: a tile above the block diagonal is the transpose of its mirror image and is never materialised).
    // Block step on pivots kb..kb+BK-1:
    //   1. the panel M[:, block] goes tiles -> exchange buffer -> row-per-lane registers p[BK] (rows above the pivot tile
    //      row come out of the pivot tile ROW, transposed)
    //   2. PB <- old panel (B operand)
    //   3. in-panel scalar sweeps (v_readlane broadcasts of the pivot row):  r = 1/p_k[t];  l_i = p_i[t]*r (i != k), l_k = -r;
    //        u != t:  p_k[u] <- 0, then p_i[u] <- fma(-l_i, pivotrow[u], p_i[u]) on every lane (lane k: = pivotrow[u]*r);  p[t] <- l
    //   4. PA <- -p (A operand); rows of the block are zero in PA and PB, so the update leaves block rows / columns alone
    //   5. stored tiles:  T[R][C] <- T[R][C] + PA_R * PB_C^T  (BK/4 v_mfma_f64_16x16x4_f64 each; the instruction is a
    //      k-ascending fma chain — verified on gfx950, tests/experiments/mfma_f64_probe.hip — so every entry receives
    //      fma(-p_i[t], old_j[t], m_ij) for t ascending, which is what the CPU checker of the test suite restates)
    //   6. write-back: M[:, block] <- p into the pivot tile column, then M[block, :] <- p^T into the pivot tile row
    // Finally the tiles are converted to the mat-vec layout a[] (see the member's comment).
    // kcol(j, z) returns K(lane, j) for j != lane, needed for j <= 16*(lane/16)+15 only (z: see below); it is called 8
    // columns at a time, one group ahead of use. diag = K(lane, lane).
    // NPIV < N — CONSTRAINT-FIRST mode (round 4; boxADMM's K = [P A'; A -1/rho], rows [NPIV, N) = the constraint block): that block is diagonal, so
    // its N - NPIV pivots are swept in CLOSED FORM — sweeping pivot NPIV + j adds rho_j A_j' A_j to the primal block, turns row / column NPIV + j
    // into -rho_j A_j and the pivot into rho_j — and only the NPIV primal pivots go through the blocked sweep (config A: 35 instead of 56 pivots,
    // nine blocks instead of fourteen: 90 + 36 matrix-core instructions instead of 140, 35 instead of 56 dependent pivot chains). kcol then returns
    // the RAW entries: lane < NPIV: P(lane, j) for j < NPIV and A(j - NPIV, lane) beyond (every primal lane: they are the operands of the rank-m
    // update); lane >= NPIV: A(lane - NPIV, j) for j < NPIV, 0 beyond. diag = P(lane, lane) / rho_lane; rho_self = this constraint lane's rho.
    // The CPU restatement of the test suite (PIVOT_SWEEP, constraint-first) forms the same matrix entry by entry and sweeps the same blocks.
    struct NoPre { template <class TT> __device__ __forceinline__ void operator()(TT&, double*, double*, int, int, int) const {} };
    // max_i |M(i, i)| over the rows i < NLIVE of the staged tiles: entry (16R + lc, 16R + lc) sits in component lc / 4 of the lanes with lc = lr + 4 (lc / 4)
    template <int NLIVE>
    __device__ __forceinline__ static double diag_abs_max(const d4 (&T)[NT][NT], int lr, int lc) {
        double dm = 0.0;
#pragma unroll
        for (int R = 0; R < (NLIVE + 15) / 16; ++R)
#pragma unroll
            for (int r = 0; r < 4; ++r) dm = fmax(dm, (lc == lr + 4 * r && 16 * R + lc < NLIVE) ? fabs(T[R][R][r]) : 0.0);
        return wave_max(dm);
    }
    // T(a, b) <- fma(rho_j A(j, a), A(j, b), T(a, b)) over MM rows of A, j ascending in groups of four (one k-step of the matrix cores each) on every
    // stored tile — the condensed register kernel (pmpc_qp_cond.hpp, at most 64 variables) calls this through invert's `pre` hook between the staging of
    // H + diag and the blocked sweep. aload(j, z): A(j, lane) (0.0 on lanes >= N); rho_of(j): rho_j, wave-uniform.
    template <int MM, class ALoad, class RhoOf>
    __device__ __forceinline__ static void rank_update(d4 (&T)[NT][NT], int ln, int lr, int lc, double* PA, double* PB, ALoad aload, RhoOf rho_of) {
        constexpr int NGR = (MM + 3) / 4, GB = 4;
        int z = 0;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int g0 = 0; g0 < NGR; g0 += GB) {
            double av_[GB * 4];
#pragma unroll
            for (int u = 0; u < GB * 4; ++u) { const int j = 4 * g0 + u; av_[u] = (j < MM) ? aload(j < MM ? j : 0, z) : 0.0; }
            sched_fence();
#pragma unroll
            for (int gg = 0; gg < GB; ++gg) {
                if (g0 + gg < NGR) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int j = 4 * (g0 + gg) + t;
                        const double rj = (j < MM) ? rho_of(j < MM ? j : 0) : 0.0;
                        PB[t * SK + ln] = av_[gg * 4 + t];
                        PA[t * SK + ln] = rj * av_[gg * 4 + t];
                    }
                    lds_order();
                    double av[NT], bv[NT];
#pragma unroll
                    for (int R = 0; R < NT; ++R) { av[R] = PA[lr * SK + 16 * R + lc]; bv[R] = PB[lr * SK + 16 * R + lc]; }
                    // a row of A touches the state columns of its segment and its own node's block only: most (group, tile) operands are all zeros, and
                    // fma(0, b, c) = c for finite b — those products are skipped (wave-uniform tests) unless an operand of the group is not finite
                    bool nz[NT]; double probe = 0.0;
#pragma unroll
                    for (int R = 0; R < NT; ++R) { nz[R] = __builtin_amdgcn_ballot_w64(bv[R] != 0.0) != 0; probe += (av[R] - av[R]) + (bv[R] - bv[R]); }
                    const bool all = __builtin_amdgcn_ballot_w64(probe != 0.0) != 0;
#pragma unroll
                    for (int R = 0; R < NT; ++R)
#pragma unroll
                        for (int C = 0; C <= R; ++C)
                            if (all || (nz[R] && nz[C])) T[R][C] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[R], bv[C], T[R][C], 0, 0, 0);
                    sched_fence();
                    lds_order();
                }
            }
        }
    }
    // EST (conditioning gate, PMPC_FLAG_ILLCOND): returns whether  max_i S_ii * max_i |(S^-1)_ii| > PMPC_COND_GATE  for the matrix S the NPIV pivots sweep
    // (within a factor of two of max S_ii / min |pivot|, a factor 2 .. 10 below cond(S) on the benchmark streams and their unbounded variants) — and then
    // returns BEFORE the conversion to the mat-vec layout: the caller gives the QP up. Costs nothing inside the sweep: the first maximum is taken from the
    // staged tiles and parked in a free LDS slot, the second from the swept tiles.
    template <int NPIV = N, bool EST = false, class KCol, class Pre = NoPre>
    __device__ __forceinline__ bool invert(int ln_in, double* st, double diag, KCol kcol, long long* tm = nullptr, double rho_self = 0.0, Pre pre = Pre(), const double gate = PMPC_COND_GATE) {
        constexpr bool CF = NPIV < N;
        constexpr int NBP = (NPIV + BK - 1) / BK;     // blocks of swept pivots
        long long tq0 = tm ? clock64() : 0;
        int ln = ln_in;
        asm volatile("" : "+v"(ln));   // keep the lane predicates below local to this function (no hoisting into long-lived SGPR masks)
        double* PA = st;
        double* PB = st + BK * SK;
        double* X = PB;
        const int lr = ln >> 4, lc = ln & 15;
        d4 T[NT][NT];   // only C <= R is used
        // row layout -> accumulator tiles, 8 columns at a time through X.
        // All N row entries are requested in ONE batch (the workspace rows come from L2 / HBM: a round trip per group of 8
        // columns cost 7 x ~2.5k cycles); they land in a[], which holds nothing live while the inverse is rebuilt.
        // `z` is an opaque zero that kcol adds to its addresses: it keeps the loads (read-only kernel arguments) and their
        // address arithmetic at this point instead of hoisted to the kernel prologue.
        // K(lane, lane) = diag replaces the loaded entry once the tiles are staged. It travels through LDS (slot 64*SX + lane
        // of X, beyond the exchange rows) and is written BEFORE the loads are issued: held in a register it was spilled,
        // and a scratch reload waits on vmcnt behind every outstanding load.
        X[64 * SX + ln] = diag;
        sched_fence();
        int z = 0;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int j = 0; j < NP; ++j) a[j] = (j < N) ? kcol(j < N ? j : 0, z) : 0.0;
        sched_fence();
        const bool isPl = ln < NPIV;
        if constexpr (CF) {   // constraint lanes: row NPIV + j of the swept matrix is -rho_j A_j
#pragma unroll
            for (int j = 0; j < NPIV; ++j) { const double sc = -(rho_self * a[j]); a[j] = isPl ? a[j] : sc; }
        }
#pragma unroll
        for (int g = 0; g < NP / SG; ++g) {
#pragma unroll
            for (int t = 0; t < SG; ++t) {
                double v = a[g * SG + t];
                if constexpr (CF) {   // primal lanes, constraint columns that share a tile with primal rows (the only ones ever read from a primal lane): -rho_j A(j, lane); a[] keeps the raw entry
                    if (g * SG + t >= NPIV && g * SG + t < N && 16 * ((g * SG + t) / 16) < NPIV) { const double sc = -(bcast_lane(rho_self, (g * SG + t < N) ? g * SG + t : 0) * v); v = isPl ? sc : v; }
                }
                X[ln * SX + t] = v;
            }
            lds_order();
            if ((lc >> 3) == (g % 2)) {
#pragma unroll
                for (int R = g / 2; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[R][g / 2][r] = X[(16 * R + lr + 4 * r) * SX + (lc & 7)];
            }
            lds_order();
            sched_fence();
        }
        // diagonal patch: entry (16R + lc, 16R + lc) of tile (R, R) sits in component lc/4 of the lanes with lc = lr + 4*(lc/4)
#pragma unroll
        for (int R = 0; R < NT; ++R) {
            const double dR = X[64 * SX + 16 * R + lc];
#pragma unroll
            for (int r = 0; r < 4; ++r) T[R][R][r] = (lc == lr + 4 * r) ? dR : T[R][R][r];
        }
        lds_order();
        if constexpr (CF) {   // rank-(N - NPIV) update of the primal tiles: T(a, b) <- fma(rho_j A(j, a), A(j, b), T(a, b)), j ascending (one k-step of the matrix cores per 4 constraints)
            constexpr int MC = N - NPIV, KS = (MC + 3) / 4, NTP = (NPIV + 15) / 16;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int j = 4 * s2 + t;
                    const double raw = (j < MC) ? a[(NPIV + j < NP) ? NPIV + j : 0] : 0.0;
                    const double rj = (j < MC) ? bcast_lane(rho_self, (NPIV + j < N) ? NPIV + j : 0) : 0.0;
                    const double pa = rj * raw;
                    PA[t * SK + ln] = (isPl && j < MC) ? pa : 0.0;
                    PB[t * SK + ln] = (isPl && j < MC) ? raw : 0.0;
                }
                lds_order();
                double av[NTP], bv[NTP];
#pragma unroll
                for (int R = 0; R < NTP; ++R) { av[R] = PA[lr * SK + 16 * R + lc]; bv[R] = PB[lr * SK + 16 * R + lc]; }
#pragma unroll
                for (int R = 0; R < NTP; ++R)
#pragma unroll
                    for (int C = 0; C <= R; ++C) T[R][C] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[R], bv[C], T[R][C], 0, 0, 0);
                sched_fence();
                lds_order();
            }
        }
        pre(T, PA, PB, ln, lr, lc);
        if constexpr (EST) { X[64 * SX + ln] = diag_abs_max<NPIV>(T, lr, lc); lds_order(); }   // (the diagonal slots are free between the patch above and the final conversion)
        if (tm) { long long t = clock64(); tm[0] += t - tq0; tq0 = t; }
#pragma unroll
        for (int b = 0; b < NBP; ++b) {
            const int kb = b * BK;
            const int w = (NPIV - kb < BK) ? NPIV - kb : BK;   // pivots of this block (a tail block of the constraint-first mode has fewer than BK)
            const int Cb = kb / 16, hb = (kb % 16) / BK;
            constexpr int RPB = BK / 4;   // accumulator registers (tile-local row groups of 4) per block
            // 1. panel -> row-per-lane registers: rows of tile rows >= Cb from tile column Cb (tile-local columns
            //    [BK*hb, BK*hb+BK)), rows of tile rows < Cb from tile row Cb (tile-local rows [BK*hb, BK*hb+BK), transposed)
            if ((lc / BK) == hb) {
#pragma unroll
                for (int R = Cb; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(16 * R + lr + 4 * r) * SX + (lc % BK)] = T[R][Cb][r];
            }
#pragma unroll
            for (int C = 0; C < Cb; ++C)
#pragma unroll
                for (int rr = 0; rr < RPB; ++rr) X[(16 * C + lc) * SX + lr + 4 * rr] = T[Cb][C][RPB * hb + rr];
            lds_order();
            double p[BK];
#pragma unroll
            for (int t = 0; t < BK; ++t) p[t] = X[ln * SX + t];
            lds_order();
            const bool inb = (ln / BK) == b && (!CF || ln < NPIV);   // (constraint-first tail block: the lanes behind the last primal pivot are ordinary rows)
            if (tm) { long long t = clock64(); tm[1] += t - tq0; tq0 = t; }
            // 2. B operand: the panel as it was at the start of the block
#pragma unroll
            for (int t = 0; t < BK; ++t) PB[t * SK + ln] = (inb || kb + t >= NPIV) ? 0.0 : p[t];
            // 3. in-panel sweeps
#pragma unroll
            for (int t = 0; t < BK; ++t) {
                const int k = kb + t;
                if (k < NPIV) {
                    const double dk = bcast_lane(p[t], k);
                    const double r = recip_uniform(dk);
                    double rk[BK];
#pragma unroll
                    for (int u = 0; u < BK; ++u) rk[u] = (u != t && u < w) ? bcast_lane(p[u], k) : 0.0;
                    double l = p[t] * r;
                    // lane k: l = -r, and its other columns start from zero, so that one fma serves every lane
                    if (w == BK) {
                        if constexpr (BK == 8) pivot_lane_setup(p[(t + 1) & 7], p[(t + 2) & 7], p[(t + 3) & 7], p[(t + 4) & 7], p[(t + 5) & 7], p[(t + 6) & 7], p[(t + 7) & 7], l, -r, k);
                        else pivot_lane_setup(p[(t + 1) & 3], p[(t + 2) & 3], p[(t + 3) & 3], l, -r, k);
                    } else {   // tail block: only the swept columns of the panel take part (the others are ordinary columns of the trailing update)
#pragma unroll
                        for (int u = 0; u < BK; ++u) if (u != t && u < w) p[u] = (ln == k) ? 0.0 : p[u];
                        l = (ln == k) ? -r : l;
                    }
#pragma unroll
                    for (int u = 0; u < BK; ++u)
                        if (u != t && u < w) p[u] = fma(-l, rk[u], p[u]);
                    p[t] = l;
                    sched_fence();
                }
            }
            if (tm) { long long t = clock64(); tm[2] += t - tq0; tq0 = t; }
            // 4. A operand
#pragma unroll
            for (int t = 0; t < BK; ++t) PA[t * SK + ln] = (inb || kb + t >= NPIV) ? 0.0 : -p[t];
            lds_order();
            // 5. rank-BK update of every stored tile
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
            if (tm) { long long t = clock64(); tm[3] += t - tq0; tq0 = t; }
            // 6. write-back of the swept panel: pivot tile column, then pivot tile row (the diagonal block ends up as the transpose of p)
#pragma unroll
            for (int t = 0; t < BK; ++t) X[ln * SX + t] = p[t];
            lds_order();
            if ((lc / BK) == hb && (lc % BK) < w) {
#pragma unroll
                for (int R = Cb; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[R][Cb][r] = X[(16 * R + lr + 4 * r) * SX + (lc % BK)];
            }
#pragma unroll
            for (int C = 0; C <= Cb; ++C)
#pragma unroll
                for (int rr = 0; rr < RPB; ++rr) {
                    const double xv_ = X[(16 * C + lc) * SX + lr + 4 * rr];
                    T[Cb][C][RPB * hb + rr] = (w == BK || lr + 4 * rr < w) ? xv_ : T[Cb][C][RPB * hb + rr];
                }
            lds_order();
            sched_fence();
        }
        if (tm) { long long t = clock64(); tm[1] += t - tq0; tq0 = t; }
        if constexpr (EST) {
            const double smax = X[64 * SX + ln];
            const double wmax = diag_abs_max<NPIV>(T, lr, lc);
            if (__builtin_amdgcn_readfirstlane((int)(smax * wmax > gate))) return true;
        }
        // accumulator tiles -> mat-vec layout, one tile row q at a time through Y (rows 16q..16q+15, all columns; blocks right of
        // the diagonal come from the mirror tiles, transposed). Columns >= N (never-consumed padding that may hold anything)
        // become exact zeros so that they drop out of the mat-vec.
        double* Y = st;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int C = 0; C < NT; ++C) {
                if (C <= q) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Y[(lr + 4 * r) * SY + 16 * C + lc] = T[q][C][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Y[lc * SY + 16 * C + lr + 4 * r] = T[C][q][r];
                }
            }
            lds_order();
            const int yo = lc * SY + 16 * (lr < NT ? lr : 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double y = Y[yo + j];
                a[16 * q + j] = (16 * (NT - 1) + j < N && NT == 4) ? y : ((16 * lr + j < N) ? y : 0.0);
            }
            lds_order();
            sched_fence();
        }
        if (tm) { long long t = clock64(); tm[4] += t - tq0; tq0 = t; }
        return false;
    }

    // K^{-1} c, one entry per lane (c = 0 on lanes >= N). Lane 16r+c forms, for each tile row q, the partial sum
    //   P_r(16q+c) = sum_j W(16q+c, 16r+j) * c_{16r+j}   (j ascending, one fma chain per q, the operand broadcast inside its row)
    // and a two-step exchange (half-waves, then neighbouring rows) adds the four partial sums of every output row and leaves
    // row i on lane i:   x_i = -((P_0 + P_2) + (P_1 + P_3)).
    __device__ __forceinline__ double apply(double c) const {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        asm volatile("s_nop 1" : "+v"(c));   // VALU write -> DPP read of the same register: 2 wait states (inline asm is not covered by the hazard recogniser)
        unroll_j<0>(acc, c);
        asm volatile("s_nop 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));   // VALU write -> v_permlane*_swap read: 2 wait states
        swap32(acc[0], acc[2]);
        swap32(acc[1], acc[3]);
        double s0 = acc[0] + acc[2], s1 = acc[1] + acc[3];
        swap16(s0, s1);
        return -(s0 + s1);
    }
    template <int J>
    __device__ __forceinline__ void unroll_j(double (&acc)[4], double c) const {
        if constexpr (J < 16) {
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = fmac_rowbcast<J>(acc[q], c, a[16 * q + J]);
            unroll_j<J + 1>(acc, c);
        }
    }
};

// boxADMM::solve_impl for compile-time (NN, MM); h/Alb/Aub/xlb/xub/x0/y0: LDS or HBM pointers; result -> out_x (NN), out_y (MM+NN)
// tr: LDS staging of RegKkt<NN+MM>::TRI doubles.
// STACKED: H and A are the upper / lower block of ONE (n+m) x n column-major array (leading dimension n+m, A = H + n):
// every lane then reads row `lane` of that array with a compile-time stride, i.e. one base address + immediate offsets.
// SYMLOWER: the Hessian block of K is taken from the LOWER triangle of H only — K(i, j) = H(max(i,j), min(i,j)) — as Eigen::LDLT reads it
// (helpers.hpp:38-43 selects the lower triangle). It matters only for a Hessian that is not bitwise symmetric: the block BFGS forms
// (-c v_i) v_j per entry (continuous_ocp.hpp:2304-2431), which differs from its mirror image in the last bit; the dense BFGS and the
// exact Hessian are bitwise symmetric, and their kernels skip the per-load select.
// GATE: the numeric conditioning gate (RegKkt::invert, EST) — the QP entry point's kernels. The fused SQP kernels decide from the bounds, once per instance
// (sqp_kernel, pmpc_launch.hpp): the gate's two diagonal extractions cost the headline kernel 54 spilled registers and 7 % (measured, same box).
template <int NN, int MM, bool STACKED = false, bool SYMLOWER = false, bool GATE = false>
__device__ __forceinline__ void boxadmm_solve_reg(const double* __restrict__ H, const double* h, const double* __restrict__ A,
                                                  const double* Alb, const double* Aub, const double* xlb, const double* xub,
                                                  const double* x0, const double* y0, const pmpc_qp_settings& s, pmpc_qp_info& info,
                                                  double* out_x, double* out_y, double* tr, long long* dbg = nullptr, long long* tm = nullptr) {
    constexpr int N = NN + MM;
    static_assert(N <= WAVE, "register-resident path needs n+m <= 64");
    const long long tp0 = dbg ? clock64() : 0;   // (phase-profile instantiation only: dbg[17] prologue, dbg[9] ADMM updates)
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
    // STACKED: the per-lane base offset goes through an empty asm statement AFTER the opaque zero is added — otherwise the
    // sum is reassociated to (roff + j*N) + zo, the loop-invariant halves are hoisted for every j and spilled (their
    // scratch reloads then sit between the loads and serialise them on vmcnt)
    // They are derived from a lane id that is re-materialised next to the loads (two v_mbcnt, tied to the opaque zero):
    // a long-lived per-lane offset is spilled, and its scratch reload in the middle of a batch of loads waits on vmcnt
    // for every load issued before it.
    auto lane_near = [](int zo) -> unsigned { unsigned l; asm("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l) : "v"(zo)); return l; };
    auto Krow = [&](int j, int zo) -> double {   // (H or A)(row of this lane, j), j < NN
        if constexpr (STACKED) { const unsigned l = lane_near(zo); unsigned b = (l < (unsigned)N ? l : 0u) + (unsigned)zo; asm("" : "+v"(b)); return H[b + (unsigned)(j * N)]; }
        else return rowp[(size_t)j * (size_t)(unsigned)(rstride + zo)];
    };
    auto KrowLower = [&](int j, int zo) -> double {   // H(max(lane, j), min(lane, j)) on primal lanes, A(row, j) on constraint lanes; j < NN
        if constexpr (STACKED) {
            const unsigned l = lane_near(zo);
            const unsigned lc = (l < (unsigned)N ? l : 0u);
            unsigned b = ((lc < (unsigned)j) ? ((unsigned)j + lc * (unsigned)N) : (lc + (unsigned)(j * N))) + (unsigned)zo;
            asm("" : "+v"(b));
            return H[b];
        } else {
            const bool up = isP && ln < j;   // one load from a selected address (H(j, lane) sits at lane * NN + j)
            const double* src = up ? H + ((size_t)lp * NN + (size_t)(unsigned)(j + zo)) : rowp + (size_t)j * (size_t)(unsigned)(rstride + zo);
            return *src;
        }
    };
    auto Acol = [&](int k, int zo) -> double {   // A(k, lane), k < MM (primal lanes)
        if constexpr (STACKED) { const unsigned l = lane_near(zo); unsigned b = (l < (unsigned)NN ? l : 0u) * N + NN + (unsigned)zo; asm("" : "+v"(b)); return H[b + (unsigned)k]; }
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
        kdiag = isP ? kd : rhov;   // (constraint lanes: the closed-form sweep leaves rho on the diagonal)
    }

    RegKkt<N> K;
    int status = PMPC_QP_UNSOLVED;
    constexpr int GAVE_UP = 100;   // internal status: the conditioning gate tripped at a factorisation (reported as UNSOLVED + PMPC_FLAG_ILLCOND)
    const double alpha = s.alpha;
    double max_Ax_z_norm = 0.0, max_Hx_ATy_h_norm = 0.0, res_prim = 1.0, res_dual = 1.0, rho_estimate = 0.0;
    // Outer loop = one KKT factorisation (first pass and after every accepted rho update); inner loop = ADMM iterations
    // on that factor. Keeping the high-register-pressure factorisation OUT of the inner loop keeps the inner loop's
    // state in registers (no scratch traffic inside the ADMM iterations).
    int iter = 1;
    int until_check = s.check_termination, until_adapt = s.adaptive_rho_interval;   // iterations left until the next multiple
    bool running = true;
    if (dbg) dbg[17] += clock64() - tp0;
    while (running) {
        {   // construct_kkt_matrix + factorise_kkt_matrix
            const long long f0 = dbg ? clock64() : 0;
            // constraint-first mode of RegKkt (round 4): the RAW entries of row `lane` of [H  A^T ; A  .] (construct_kkt_matrix, box_admm.hpp:209-223);
            // the diagonal constraint block is swept in closed form inside invert (kdiag carries rho on the constraint lanes)
            // Conditioning gate (PMPC_FLAG_ILLCOND, include/polympc_amd.h). The constraint-first sweep inverts S = P + A' diag(rho) A inside K; cond(S) =
            // rho_eq |A|^2 / lambda_min(P on null A) stays ~1e5 whatever rho is while the directions A leaves free are bounded variables (rho_box scales
            // with rho: every BASELINE workload — there this order is MORE accurate than the reference's pivoted LDL^T, tests/test_oracle_pins.py), and
            // grows with rho when unbounded variables (rho_box = RHO_MIN) span them. Beyond the gate (RegKkt::invert, EST) the QP is given up (UNSOLVED +
            // the flag) and the launcher's redo launch solves it in the full KKT form (LDS-resident static LDL^T). Wave-uniform; restated by the CPU checker.
            // (A second, full-sweep instantiation of invert() as an in-kernel fallback cost the headline kernel 87 spilled registers; a per-pivot
            // running minimum inside the sweep 60 more SGPR spills kernel-wide, +6 % on the bench line: both dropped.)
            const bool tripped = K.template invert<NN, GATE>(ln, tr, kdiag, [&](int j, int z) -> double {
                if (j < NN) { if constexpr (SYMLOWER) return KrowLower(j < NN ? j : 0, z); else return Krow(j < NN ? j : 0, z); }
                const double v = Acol(j >= NN ? j - NN : 0, z);   // every primal lane: column `lane` of A is its operand of the rank-m update
                if constexpr (STACKED) return lane_near(z) < (unsigned)NN ? v : 0.0;
                return isP ? v : 0.0;
            }, tm, rhov);
            if constexpr (GATE) { if (tripped) { status = GAVE_UP; running = false; if (dbg) dbg[0] += clock64() - f0; break; } }
            if (dbg) dbg[0] += clock64() - f0;
        }
        bool refactor = false;
        for (; iter <= s.max_iter; ++iter) {
            const long long ti0 = dbg ? clock64() : 0;
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
            if (dbg) dbg[9] += clock64() - ti0;
            // iter % check_termination == 0 / iter % adaptive_rho_interval == 0 (box_admm.hpp:141,:160) as countdowns
            bool check = false, adapt = false;
            if (s.check_termination != 0 && --until_check == 0) { check = true; until_check = s.check_termination; }
            if (s.adaptive_rho && --until_adapt == 0) { adapt = true; until_adapt = s.adaptive_rho_interval; }
            if (check || adapt) {  // residuals_update, box_admm.hpp:398-415
                const long long r0 = dbg ? clock64() : 0;
                // loads in chunks of RC columns (independent, coalesced), each followed by its slice of the mat-vec chain
                // H / A row and A column in ONE list of NN + MM entries, RC at a time: the loads are L2 round trips (~2 k cycles each batch) and
                // the two mat-vec chains keep their ascending order whatever the batching
                constexpr int RC = PMPC_REG_RESIDUAL_BATCH;
                int zr = 0;            // opaque zero added to the addresses: keeps these loop-invariant loads inside the loop
                asm volatile("" : "+v"(zr));
                double acc = 0.0;      // lanes < n: (H x)_i ; lanes in [n, N): (A x)_r
                double aty = 0.0;      // lanes < n: (A^T y_a)_i
#pragma unroll
                for (int e0 = 0; e0 < N; e0 += RC) {
                    double mv[RC];
#pragma unroll
                    for (int e = 0; e < RC; ++e) {
                        const int ee = e0 + e;
                        mv[e] = (ee < NN) ? Krow(ee < NN ? ee : 0, zr) : ((ee < N) ? Acol((ee >= NN && ee < N) ? ee - NN : 0, zr) : 0.0);
                    }
#pragma unroll
                    for (int e = 0; e < RC; ++e) {
                        const int ee = e0 + e;
                        if (ee < NN) acc += mv[e] * bcast_lane(xv, ee);
                        else if (ee < N) aty += mv[e] * bcast_lane(yv, ee);
                    }
                    sched_fence();
                }
                // max is exact and order-free: the maximum of several infinity norms is ONE wave reduction of the per-lane maxima
                //   max(|Ax|, |z|, |x|):  constraint lanes carry |(Ax)_r| and |z_r|, primal lanes |x_i|
                //   max(|Hx|, |A'y|, |h|, |y_box|):  primal lanes only (hv is zero elsewhere)
                const double ax = fabs(xv);
                max_Ax_z_norm = wave_max(isC ? fmax(fabs(acc), ax) : (isP ? ax : 0.0));
                max_Hx_ATy_h_norm = wave_max(isP ? fmax(fmax(fabs(acc), fabs(aty)), fmax(fabs(hv), fabs(yv))) : 0.0);
                const double rp = wave_max(isC ? fabs(acc - xv) : 0.0), rq = wave_max(isP ? fabs(xv - qv) : 0.0);
                res_prim = rp + rq;
                res_dual = wave_max(isP ? fabs(((acc + hv) + aty) + yv) : 0.0);
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
                    kdiag = isP ? (kdiag + (rhov - prev)) : rhov;   // update_kkt_rho, box_admm.hpp:448-452
                    refactor = true;
                    ++iter;
                    break;
                }
            }
        }
        if (!refactor) running = false;
    }
    const bool gave_up = GATE && status == GAVE_UP;
    if (gave_up) status = PMPC_QP_UNSOLVED;
    else if (iter > s.max_iter) status = PMPC_QP_MAX_ITER_EXCEEDED;
    if (isP) { out_x[ln] = xv; out_y[MM + ln] = yv; }
    if (isC) out_y[r] = yv;
    const bool bad = __builtin_amdgcn_ballot_w64(((xv - xv) + (yv - yv)) != 0.0) != 0;   // non-finite x or y on any lane
    info.status = status; info.iter = iter; info.rho_updates = rho_updates; info.flags = (bad ? PMPC_FLAG_NONFINITE : 0) | (gave_up ? PMPC_FLAG_ILLCOND : 0);
    info.rho_estimate = rho_estimate; info.res_prim = res_prim; info.res_dual = res_dual;
}

}  // namespace pmpc
