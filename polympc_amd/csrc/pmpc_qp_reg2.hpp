// polympc_amd — register-resident box-ADMM QP solve for compile-time sizes with 64 < n+m <= 112 (one wavefront per QP, TWO KKT rows
// per lane: rows `lane` and `lane + 64`). Config B (CSTR, 11 nodes: n = 66, m = 44, 110 rows) and the reference's own 11-node robot
// grid (88 rows) run here instead of on the LDS-resident LDL^T path, whose 2N-step substitution chain per ADMM iteration and N^2/2
// LDS-resident factor (49 KB at 110 rows: two instances per CU) bound those configurations.
//
// Same algorithm, constants and update order as pmpc_qp.hpp / pmpc_qp_reg.hpp (boxADMM::solve_impl, box_admm.hpp:88-205). The linear
// algebra is the blocked symmetric sweep of pmpc_qp_reg.hpp — W = -K^{-1} in 16x16 fp64 MFMA accumulator tiles, 4 pivots per block,
// one v_mfma_f64_16x16x4_f64 per stored tile and block — carried to 7 x 7 tiles, with one difference in how W is APPLIED:
//   * the accumulator tiles ARE the mat-vec operand. Tile (R, C) holds M(16R + lr + 4r, 16C + lc) on lane (lr, lc) = (lane/16, lane%16),
//     register r. Read through the symmetry of W as "row 16C + lc, column 16R + 4r + lr", lane (lr, lc) owns, for each of the NT
//     output rows 16C + lc, the 4 NT columns j = lr (mod 4): x = -W rhs is NT independent fma chains per lane (one per tile column C)
//     over (R, r) ascending, the rhs entry broadcast inside the 16-lane row by the DPP modifier of v_fmac_f64, followed by the
//     reduce-scatter of pmpc_qp_reg.hpp over the four 16-lane rows ((P0 + P2) + (P1 + P3)), twice (rows < 64, rows >= 64).
//     No conversion pass, no second copy of W: after the sweep the 21 tiles above the block diagonal are filled with the transposes
//     of their mirror images and the diagonal tiles are transposed in place (through LDS, once per factorisation).
//   * the rhs reaches the broadcast layout — lane (lr, k) of operand register s holds entry 64 s + 16 (k / 4) + 4 (k % 4) + lr —
//     by one LDS round trip per iteration (two stores, two loads).
// Registers: 49 tiles = 392 of the 512 (arch + accumulation) registers of a wavefront that owns its SIMD (launch bound: 1 wave / SIMD);
// the compiler keeps part of the tiles in AGPRs and reads them back (v_accvgpr_read) next to the fma that consumes them.
// The CPU restatement of exactly this order is PIVOT_SWEEP2 (tests: bit for bit).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "pmpc_qp_reg.hpp"
#include "pmpc_jview.hpp"

namespace pmpc {

template <int N, int NVREQ = -1>   // NVREQ >= 0: operand tiles kept in arch VGPRs (see NV)
struct RegKkt2 {
    static_assert(N > 64 && N <= 128, "two-rows-per-lane register path: 65..128 KKT rows");
    using d4 = double __attribute__((ext_vector_type(4)));
    static constexpr int BK = 4;
    static constexpr int NT = (N + 15) / 16;          // 16x16 tiles per dimension (5..7)
    static constexpr int NP = NT * 16;
    static constexpr int NB = (N + BK - 1) / BK;
    static constexpr int SG = 8;                      // columns per group of the row -> tile staging
    static constexpr int SK = 144;                    // k-major operand panels: (t, row) at t*SK + row, 128 row slots (every lane stores both of
                                                      // its rows unconditionally), 16 (mod 32): conflict-free MFMA operand reads
    static constexpr int SX = SG + 1;                 // row-major exchange buffer: (row, t) at row*SX + t, 128 rows + 128 diagonal slots
    static constexpr int XSZ = 128 * SX + 128;
    static constexpr int SY = 17;                     // transposition buffer: one 16x16 tile, row stride 17
    static constexpr int TRI0 = BK * SK + (XSZ > BK * SK ? XSZ : BK * SK);
    // 113..128 rows (8 x 8 tiles): the 64 operand tiles of the mat-vec are 512 registers — the whole register file. Sixteen of them — two per tile
    // row: the tiles with (R + C) mod 4 = 3 — live in LDS instead, lane-major (tile l, component pair h, lane: two doubles at l*256 + h*128 + 2*lane,
    // one conflict-free 16-byte read per lane), and are read two steps ahead of the fma that consumes them; the other 48 are placed as in the 7 x 7
    // case (17 in VGPRs, 31 in the accumulation file). Spread over the tile rows like this every step of the mat-vec has the same mix (six register
    // operands, two from LDS). The sweep itself needs the 36 lower tiles only, all in registers. The LDS tiles alias the sweep's staging (dead once the
    // inverse is finished); a one-tile transposition buffer behind them also serves as the per-iteration rhs buffer.
    static constexpr bool LDS_TILES = NT == 8;
    static constexpr int NL = LDS_TILES ? 2 * NT : 0;
    static constexpr int LT = NL * 256;                       // doubles of LDS tiles; the small buffer YS sits behind them
    static constexpr int YS = 16 * SY;
    static constexpr int TRI1 = TRI0 > NT * 16 * SY ? TRI0 : NT * 16 * SY;
#ifndef PMPC_REG2_LDSPARK
#define PMPC_REG2_LDSPARK 9   /* operand tiles parked in LDS while the residuals are evaluated (config B: 0 / 6 / 9 -> 38.4 / 37.0 / ... ms), see park() */
#endif
    // 7 x 7 tiles: room for three more parked tiles behind the staging (6 KB: the largest of these kernels then needs 39.3 KB, four instances per CU still fit)
    static constexpr int PARK_EXTRA = (NT == 7 && PMPC_REG2_LDSPARK > 6) ? (PMPC_REG2_LDSPARK - 6) * 256 : 0;
    static constexpr int TRI = LDS_TILES ? (TRI0 > LT + YS ? TRI0 : LT + YS) : TRI1 + PARK_EXTRA;   // doubles of LDS staging (PA | PB aliased with X; Y and the rhs buffer alias both)
    static constexpr int RHS_OFF = LDS_TILES ? LT : 0;        // where apply() stages the right-hand side / the residual evaluation its vectors
    __device__ __forceinline__ static constexpr bool in_lds(int R, int C) { return LDS_TILES && ((R + C) & 3) == 3; }
    __device__ __forceinline__ static constexpr int lds_index(int R, int C) { return 2 * R + (C >= 4 ? 1 : 0); }
    static constexpr int NTR = LDS_TILES ? NT - 2 : NT;       // register-resident operand tiles per tile row

    d4 T[NT][NT];   // after invert(): T[R][C] = the operand tile of output rows 16C + lc against columns 16R + 4r + lr
    // Register-file placement of the finished operand tiles, by hand: 49 tiles are 392 registers, more than either file holds (256 each),
    // and v_fmac_f64 reads arch VGPRs only. Left to the allocator, most of W went to SCRATCH (162 of 196 doubles reloaded per ADMM iteration).
    // Tiles with index R*NT + C < NV stay in arch VGPRs; the others are split into 32-bit halves whose only uses are "a"-constrained inline-asm
    // operands, which makes their virtual registers AGPR-class: they live in the accumulation file for the whole ADMM loop and are copied
    // (v_accvgpr_read_b32 x 2) into a temporary pair next to the fma that consumes them.
    static constexpr int NV = NVREQ >= 0 ? NVREQ : (LDS_TILES ? 17 : ((NT == 7) ? 18 : (NT == 6 ? 12 : 10)));
    static constexpr int NA = NT * NTR - NV;
    // running index of a register-resident tile, row-major over the tiles that are not in LDS
    __device__ __forceinline__ static constexpr int reg_index(int R, int C) {
        if (!LDS_TILES) return R * NT + C;
        const int c1 = (3 - R) & 3, c2 = c1 + 4;
        return R * NTR + C - (C > c1 ? 1 : 0) - (C > c2 ? 1 : 0);
    }
    __device__ __forceinline__ static constexpr bool in_agpr(int R, int C) { return !in_lds(R, C) && reg_index(R, C) >= NV; }
    int Alo[NA * 4], Ahi[NA * 4];

    // one block step of the sweep (pivots 4b .. 4b+3); a member template so that every tile index, lane index and EXEC mask below is
    // a compile-time constant (a 28-trip loop of this size is not unrolled by the optimiser on request)
    template <int b>
    __device__ __forceinline__ void block_step(int ln, int lr, int lc, double* PA, double* PB, double* X, long long* tm, long long& tq0) {
            constexpr int kb = b * BK;
            constexpr int Cb = kb / 16, hb = (kb % 16) / BK;
            // 1. panel -> row-per-lane registers (rows above the pivot tile row come out of the pivot tile ROW, transposed)
            if ((lc / BK) == hb) {
#pragma unroll
                for (int R = Cb; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(16 * R + lr + 4 * r) * SX + (lc % BK)] = T[R][Cb][r];
            }
#pragma unroll
            for (int C = 0; C < Cb; ++C) X[(16 * C + lc) * SX + lr] = T[Cb][C][hb];
            lds_order();
            double p0[BK], p1[BK];
#pragma unroll
            for (int t = 0; t < BK; ++t) { p0[t] = X[ln * SX + t]; p1[t] = X[(64 + ln) * SX + t]; }
            lds_order();
            const bool inb0 = (ln / BK) == b, inb1 = ((64 + ln) / BK) == b;
            const bool dead1 = (64 + ln) >= N;        // no second row on this lane: its registers stay exact zeros
            if (tm) { long long t = clock64(); tm[1] += t - tq0; tq0 = t; }
            // 2. B operand: the panel as it was at the start of the block (block rows and pivots >= N are zero)
#pragma unroll
            for (int t = 0; t < BK; ++t) {
                PB[t * SK + ln] = (inb0 || kb + t >= N) ? 0.0 : p0[t];
                PB[t * SK + 64 + ln] = (inb1 || dead1 || kb + t >= N) ? 0.0 : p1[t];
            }
            // 3. in-panel sweeps (pivot row broadcast with v_readlane from the register set that holds row k)
#pragma unroll
            for (int t = 0; t < BK; ++t) {
                const int k = kb + t;
                if (k < N) {
                    const double dk = (k < 64) ? bcast_lane(p0[t], k & 63) : bcast_lane(p1[t], (k - 64) & 63);
                    const double r = recip_uniform(dk);
                    double rk[BK];
#pragma unroll
                    for (int u = 0; u < BK; ++u) rk[u] = (u != t) ? ((k < 64) ? bcast_lane(p0[u], k & 63) : bcast_lane(p1[u], (k - 64) & 63)) : 0.0;
                    double l0 = p0[t] * r, l1 = p1[t] * r;
                    if (k < 64) pivot_lane_setup(p0[(t + 1) & 3], p0[(t + 2) & 3], p0[(t + 3) & 3], l0, -r, k & 63);
                    else pivot_lane_setup(p1[(t + 1) & 3], p1[(t + 2) & 3], p1[(t + 3) & 3], l1, -r, (k - 64) & 63);
#pragma unroll
                    for (int u = 0; u < BK; ++u)
                        if (u != t) { p0[u] = fma(-l0, rk[u], p0[u]); p1[u] = fma(-l1, rk[u], p1[u]); }
                    p0[t] = l0; p1[t] = l1;
                    sched_fence();
                }
            }
            if (tm) { long long t = clock64(); tm[2] += t - tq0; tq0 = t; }
            // 4. A operand
#pragma unroll
            for (int t = 0; t < BK; ++t) {
                PA[t * SK + ln] = (inb0 || kb + t >= N) ? 0.0 : -p0[t];
                PA[t * SK + 64 + ln] = (inb1 || dead1 || kb + t >= N) ? 0.0 : -p1[t];
            }
            lds_order();
            // 5. rank-4 update of every stored tile: T[R][C] += PA_R * PB_C^T (a k-ascending fma chain per entry)
            {
                double av[NT], bv[NT];
#pragma unroll
                for (int R = 0; R < NT; ++R) { av[R] = PA[lr * SK + 16 * R + lc]; bv[R] = PB[lr * SK + 16 * R + lc]; }
#pragma unroll
                for (int R = 0; R < NT; ++R)
#pragma unroll
                    for (int C = 0; C <= R; ++C) T[R][C] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[R], bv[C], T[R][C], 0, 0, 0);
                sched_fence();
            }
            lds_order();
            if (tm) { long long t = clock64(); tm[3] += t - tq0; tq0 = t; }
            // 6. write-back of the swept panel: pivot tile column, then pivot tile row
#pragma unroll
            for (int t = 0; t < BK; ++t) { X[ln * SX + t] = p0[t]; X[(64 + ln) * SX + t] = dead1 ? 0.0 : p1[t]; }
            lds_order();
            if ((lc / BK) == hb) {
#pragma unroll
                for (int R = Cb; R < NT; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[R][Cb][r] = X[(16 * R + lr + 4 * r) * SX + (lc % BK)];
            }
#pragma unroll
            for (int C = 0; C <= Cb; ++C) T[Cb][C][hb] = X[(16 * C + lc) * SX + lr];
            lds_order();
            sched_fence();
    }
    template <int b>
    __device__ __forceinline__ void block_steps(int ln, int lr, int lc, double* PA, double* PB, double* X, long long* tm, long long& tq0) {
        if constexpr (b < NB) { block_step<b>(ln, lr, lc, PA, PB, X, tm, tq0); block_steps<b + 1>(ln, lr, lc, PA, PB, X, tm, tq0); }
    }

    // kload(j, s, z): K(row 64 s + lane, j) for j != row (0.0 for rows >= N), needed for the columns of the block-lower tile storage only
    // (j < 16 (row / 16 + 1)); z is the opaque zero that keeps the address arithmetic next to the loads. diag0 / diag1: K(row, row).
    struct NoPre { __device__ __forceinline__ void operator()(RegKkt2&, double*, double*, int, int, int) const {} };
    // T[R][C] += sum_j (rho_j a_j)_R (a_j)_C' over the MM rows a_j of A, j ascending in groups of four — the k-ascending fma chain of
    // v_mfma_f64_16x16x4_f64 per stored entry: M(a, b) = fma(rho_j A(j, a), A(j, b), M(a, b)). aload(j, e, z): A(j, row 64 e + lane) (0.0 on lanes without
    // such a row); rho_of(j): rho_j, wave-uniform. The condensed register kernel (pmpc_qp_cond.hpp) calls this between the staging of H + diag and the
    // blocked sweep: the tiles then hold S = H + sigma I + rho_box + A' diag(rho) A.
    template <int MM, class ALoad, class RhoOf>
    __device__ __forceinline__ void rank_update(int ln, int lr, int lc, double* PA, double* PB, ALoad aload, RhoOf rho_of) {
        constexpr int NGR = (MM + BK - 1) / BK, GB = 4;   // groups of four rows; GB groups (32 loads) requested together
        int z = 0;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int g0 = 0; g0 < NGR; g0 += GB) {
            double a0[GB * BK], a1[GB * BK];
#pragma unroll
            for (int u = 0; u < GB * BK; ++u) {
                const int j = BK * g0 + u;
                a0[u] = (j < MM) ? aload(j < MM ? j : 0, 0, z) : 0.0;
                a1[u] = (j < MM) ? aload(j < MM ? j : 0, 1, z) : 0.0;
            }
            sched_fence();
#pragma unroll
            for (int gg = 0; gg < GB; ++gg) {
                if (g0 + gg < NGR) {
#pragma unroll
                    for (int t = 0; t < BK; ++t) {
                        const int j = BK * (g0 + gg) + t;
                        const double rj = (j < MM) ? rho_of(j < MM ? j : 0) : 0.0;
                        PB[t * SK + ln] = a0[gg * BK + t];
                        PB[t * SK + 64 + ln] = a1[gg * BK + t];
                        PA[t * SK + ln] = rj * a0[gg * BK + t];
                        PA[t * SK + 64 + ln] = rj * a1[gg * BK + t];
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
                    lds_order();
                    sched_fence();
                }
            }
        }
    }
    template <class KLoad, class Pre = NoPre>
    __device__ __forceinline__ void invert(int ln_in, double* st, double diag0, double diag1, KLoad kload, long long* tm = nullptr, Pre pre = Pre()) {
        long long tq0 = tm ? clock64() : 0;
        int ln = ln_in;
        asm volatile("" : "+v"(ln));
        double* PA = st;
        double* PB = st + BK * SK;
        double* X = PB;
        const int lr = ln >> 4, lc = ln & 15;
        X[128 * SX + ln] = diag0;
        X[128 * SX + 64 + ln] = diag1;
        sched_fence();
        int z = 0;
        asm volatile("" : "+v"(z));
        // rows -> accumulator tiles in batches of 16 columns: all loads of a batch are in flight together;
        // rows below 64 have stored tiles in the first 64 columns only
        constexpr int CB = 16;
#pragma unroll
        for (int c0 = 0; c0 < NP; c0 += CB) {
            double e0[CB], e1[CB];
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                e0[j] = (c0 < 64 && c0 + j < N) ? kload(c0 + j < N ? c0 + j : 0, 0, z) : 0.0;
                e1[j] = (c0 + j < N) ? kload(c0 + j < N ? c0 + j : 0, 1, z) : 0.0;
            }
            sched_fence();
#pragma unroll
            for (int gg = 0; gg < CB / SG; ++gg) {
                const int g = c0 / SG + gg;                // global column group; tile column g / 2, half g % 2
                if (g * SG >= NP) break;
#pragma unroll
                for (int t = 0; t < SG; ++t) { if (c0 < 64) X[ln * SX + t] = e0[gg * SG + t]; X[(64 + ln) * SX + t] = e1[gg * SG + t]; }
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
        }
#pragma unroll
        for (int R = 0; R < NT; ++R) {   // diagonal patch: entry (16R + lc, 16R + lc) sits in component lc / 4 of the lanes with lc = lr + 4 (lc / 4)
            const double dR = X[128 * SX + 16 * R + lc];
#pragma unroll
            for (int r = 0; r < 4; ++r) T[R][R][r] = (lc == lr + 4 * r) ? dR : T[R][R][r];
        }
        lds_order();
        pre(*this, PA, PB, ln, lr, lc);
        if (tm) { long long t = clock64(); tm[0] += t - tq0; tq0 = t; }
        block_steps<0>(ln, lr, lc, PA, PB, X, tm, tq0);
        if (tm) { long long t = clock64(); tm[1] += t - tq0; tq0 = t; }
        // mirror: T[C][R] <- T[R][C]^T for R > C (the tiles the block-lower storage never materialised), and T[C][C] <- T[C][C]^T: the
        // operand tile of output row block C against column block R is the transpose of W's (C, R) tile. One tile row per LDS round.
        double* Y = st;
        if constexpr (!LDS_TILES) {
#pragma unroll
            for (int R = 0; R < NT; ++R) {   // one tile row per LDS round: tiles (R, 0..R) out, their transposes back into (0..R, R)
#pragma unroll
                for (int C = 0; C <= R; ++C)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Y[C * 16 * SY + (lr + 4 * r) * SY + lc] = T[R][C][r];
                lds_order();
#pragma unroll
                for (int C = 0; C <= R; ++C)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[C][R][r] = Y[C * 16 * SY + lc * SY + lr + 4 * r];
                lds_order();
                sched_fence();
            }
        } else {
            // 8 x 8 tiles: one tile at a time through the small buffer behind the LDS tiles (everything in front of it is being filled with tiles). A lower
            // tile that lives in LDS is stored as it is, its transpose becomes the mirror tile — in registers or in LDS.
            double* Lt = st; double* Ys = st + LT;
            auto to_lds = [&](int li, const d4& v) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Lt[li * 256 + (r >> 1) * 128 + 2 * ln + (r & 1)] = v[r];
            };
            mirror_tiles<0, 0>(Ys, lr, lc, to_lds);
            lds_order();
        }
#pragma unroll
        for (int R = 0; R < NT; ++R)
#pragma unroll
            for (int C = 0; C < NT; ++C)
                if (in_agpr(R, C)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { Alo[(reg_index(R, C) - NV) * 4 + r] = __double2loint(T[R][C][r]); Ahi[(reg_index(R, C) - NV) * 4 + r] = __double2hiint(T[R][C][r]); }
                }
        if (tm) { long long t = clock64(); tm[4] += t - tq0; tq0 = t; }
    }
    template <int R, int C, class ToLds>
    __device__ __forceinline__ void mirror_tiles(double* Ys, int lr, int lc, ToLds to_lds) {
        if constexpr (R < NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Ys[(lr + 4 * r) * SY + lc] = T[R][C][r];
            if constexpr (C < R && in_lds(R, C)) to_lds(lds_index(R, C), T[R][C]);
            lds_order();
            d4 tt;
#pragma unroll
            for (int r = 0; r < 4; ++r) tt[r] = Ys[lc * SY + lr + 4 * r];
            lds_order();
            if constexpr (in_lds(C, R)) to_lds(lds_index(C, R), tt);
            else T[C][R] = tt;
            sched_fence();
            if constexpr (C < R) mirror_tiles<R, C + 1>(Ys, lr, lc, to_lds);
            else mirror_tiles<R + 1, 0>(Ys, lr, lc, to_lds);
        }
    }
    // one entry of an AGPR-resident tile -> a temporary VGPR pair (volatile: stays inside the ADMM loop, next to its use)
    __device__ __forceinline__ static double from_agpr(int alo, int ahi) {
        int tl, th;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(tl) : "a"(alo));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(th) : "a"(ahi));
        return __hiloint2double(th, tl);
    }

    // The VGPR-resident tiles leave the register file while the residuals are evaluated (every check_termination-th iteration): that code
    // wants dozens of loads in flight, and with 136 registers pinned by the operand the allocator spilled the LOADED values instead — a
    // scratch round trip between any two loads, i.e. one exposed L2 latency per matrix entry (0.37 ms per check on 4096 QPs; the ADMM
    // iteration itself takes 0.012 ms). `mem` is a private (scratch) array, indexed through an opaque zero so that it stays memory.
    // NLP of the NV tiles are parked in LDS instead (the sweep's staging is free between factorisations; k-major, one double per lane and slot:
    // conflict-free 8-byte accesses): 2 KB per tile that neither leave the CU nor come back through L2 / HBM.
    // USED: doubles at the head of the staging that the residual evaluation itself occupies (x, y, the products of the second slot)
    __device__ __forceinline__ static constexpr int lds_parked(int used) {
        if (LDS_TILES) return 0;
        const int fit = (TRI - used) / 256;
        return fit < 0 ? 0 : (fit < PMPC_REG2_LDSPARK ? fit : (PMPC_REG2_LDSPARK < NV ? PMPC_REG2_LDSPARK : NV));
    }
    static constexpr int NPARK = NV * 4;
    template <int NLP, int PARK_OFF>
    __device__ __forceinline__ void park(double* mem, int oz, double* lds, unsigned lane) {
#pragma unroll
        for (int R = 0; R < NT; ++R)
#pragma unroll
            for (int C = 0; C < NT; ++C)
                if (!in_agpr(R, C) && !in_lds(R, C)) {
                    const int ri = reg_index(R, C);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (ri < NLP) lds[PARK_OFF + (ri * 4 + r) * 64 + lane] = T[R][C][r];
                        else mem[(ri - NLP) * 4 + r + oz] = T[R][C][r];
                    }
                }
    }
    template <int NLP, int PARK_OFF>
    __device__ __forceinline__ void unpark(const double* mem, int oz, const double* lds, unsigned lane) {
#pragma unroll
        for (int R = 0; R < NT; ++R)
#pragma unroll
            for (int C = 0; C < NT; ++C)
                if (!in_agpr(R, C) && !in_lds(R, C)) {
                    const int ri = reg_index(R, C);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (ri < NLP) T[R][C][r] = lds[PARK_OFF + (ri * 4 + r) * 64 + lane];
                        else T[R][C][r] = mem[(ri - NLP) * 4 + r + oz];
                    }
                }
    }

    // K^{-1} c for the two entries of every lane: c0 = entry `lane`, c1 = entry `lane + 64` (exact zero where that is >= N).
    __device__ __forceinline__ void apply(double c0, double c1, double* st_, int ln, double& x0, double& x1) const {
        const int lr = ln >> 4, lc = ln & 15;
        double* st = st_ + RHS_OFF;
        st[ln] = c0; st[64 + ln] = c1;
        lds_order();
        const int pidx = 16 * (lc >> 2) + 4 * (lc & 3) + lr;
        double b0 = st[pidx], b1 = st[64 + pidx];
        lds_order();
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        asm volatile("s_nop 1" : "+v"(b0), "+v"(b1));   // VALU / LDS write -> DPP read: wait states inline asm is not covered for
        lt_ = reinterpret_cast<const d2*>(st_) + ln;
        d2 lpa[NT], lpb[NT];
        lds_step_pairs<0>(lpa);
        chain<0>(acc, b0, b1, lpa, lpb);
        asm volatile("s_nop 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
        swap32(acc[0], acc[2]); swap32(acc[1], acc[3]);
        double s0 = acc[0] + acc[2], s1 = acc[1] + acc[3];
        swap16(s0, s1);
        x0 = -(s0 + s1);
        swap32(acc[4], acc[6]); swap32(acc[5], acc[7]);
        double u0 = acc[4] + acc[6], u1 = acc[5] + acc[7];
        swap16(u0, u1);
        x1 = -(u0 + u1);
    }
    // step (R, r) of every chain: acc[C] = fma(T[R][C][r], rhs(16R + 4r + lr), acc[C]), C < NT; (R, r) ascending
    using d2 = double __attribute__((ext_vector_type(2)));
    mutable const d2* lt_ = nullptr;   // this lane's slot in the LDS tiles (set by apply): tile l, component pair h at lt_[l * 128 + h * 64]
    // the LDS operands of steps S, S + 1 (S even: components r, r + 1 of every LDS tile of tile row S / 4) in one 16-byte read per tile
    template <int S>
    __device__ __forceinline__ void lds_step_pairs(d2 (&lp)[NT]) const {
        if constexpr (LDS_TILES && S < 4 * NT) {
            constexpr int R = S / 4, h = (S % 4) >> 1;
#pragma unroll
            for (int C = 0; C < NT; ++C) if (in_lds(R, C)) lp[C] = lt_[lds_index(R, C) * 128 + h * 64];
        }
    }
    // component r of the operand tiles (R, C), C < NT, wherever they live
    template <int R, int r, int C>
    __device__ __forceinline__ void operands(double (&w)[NT], const d2 (&cur)[NT], int c_lo, int c_hi) const {   // (c_lo, c_hi: compile-time after inlining)
        if constexpr (C < NT) {
            if (C >= c_lo && C < c_hi) {
                if constexpr (in_lds(R, C)) w[C] = cur[C][r & 1];
                else if constexpr (in_agpr(R, C)) w[C] = from_agpr(Alo[(reg_index(R, C) - NV) * 4 + r], Ahi[(reg_index(R, C) - NV) * 4 + r]);
                else w[C] = T[R][C][r];
            }
            operands<R, r, C + 1>(w, cur, c_lo, c_hi);
        }
    }
    // cur: LDS operands of this step (read two steps ago); nxt: buffer for the pair after it, requested at the even steps before the fma are issued
    template <int S>
    __device__ __forceinline__ void chain(double (&acc)[8], double b0, double b1, d2 (&cur)[NT], d2 (&nxt)[NT]) const {
        if constexpr (S < 4 * NT) {
            constexpr int R = S / 4, r = S % 4;
            if constexpr ((r & 1) == 0) lds_step_pairs<S + 2>(nxt);
            // the operands of a step first (accumulation-file reads), then its fma: a v_accvgpr_read directly in front of the DPP operation that
            // consumes it needs a wait state (an s_nop per operand, 98 per ADMM iteration at 7 x 7 tiles)
            double w[NT];
            constexpr int G = 4;   // (operands in groups of four: with all NT of a step gathered first the loop spilled at the QP entry point)
#pragma unroll
            for (int C0 = 0; C0 < NT; C0 += G) {
                operands<R, r, 0>(w, cur, C0, C0 + G);
#pragma unroll
                for (int C = C0; C < C0 + G && C < NT; ++C) {
                    if constexpr (R < 4) acc[C] = fmac_rowbcast<4 * (R & 3) + r>(acc[C], b0, w[C]);
                    else acc[C] = fmac_rowbcast<4 * (R & 3) + r>(acc[C], b1, w[C]);
                }
            }
            if constexpr ((r & 1) == 0) chain<S + 1>(acc, b0, b1, cur, nxt);
            else chain<S + 1>(acc, b0, b1, nxt, cur);
        }
    }
};

// boxADMM::solve_impl for compile-time (NN, MM), 64 < NN + MM <= 112. Arguments as boxadmm_solve_reg; tr: RegKkt2<NN+MM>::TRI doubles of LDS.
// JV: block-sparse view of A (pmpc_jview.hpp) when the QP comes from the fused SQP kernel (STACKED) — the residual evaluation then forms A x and
// A' y from LDS instead of re-reading the dense A from the workspace; NoJView at the plain QP entry points.
template <int NN, int MM, bool STACKED = false, bool SYMLOWER = false, class JV = NoJView>
__device__ __forceinline__ void boxadmm_solve_reg2(const double* __restrict__ H, const double* h, const double* __restrict__ A,
                                                   const double* Alb, const double* Aub, const double* xlb, const double* xub,
                                                   const double* x0, const double* y0, const pmpc_qp_settings& s, pmpc_qp_info& info,
                                                   double* out_x, double* out_y, double* tr, long long* dbg = nullptr, long long* tm = nullptr,
                                                   const JV& jv = JV()) {
    constexpr int N = NN + MM;
    constexpr bool HASJ = STACKED && !std::is_same<JV, NoJView>::value;
    static_assert(N > WAVE && N <= 128, "two-rows-per-lane register path");
    const int ln = lane_id();
    // slot e: KKT row i_e = lane + 64 e. Primal rows [0, NN), constraint rows [NN, N).
    int idx[2]; bool isP[2], isC[2]; int rc[2], lp[2];
    double hv[2], lo[2], hi[2]; int type[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        idx[e] = ln + 64 * e;
        isP[e] = idx[e] < NN; isC[e] = (idx[e] >= NN) && (idx[e] < N);
        rc[e] = isC[e] ? idx[e] - NN : 0;
        lp[e] = isP[e] ? idx[e] : 0;
        hv[e] = isP[e] ? h[lp[e]] : 0.0;
        lo[e] = isP[e] ? xlb[lp[e]] : (isC[e] ? Alb[rc[e]] : 0.0);
        hi[e] = isP[e] ? xub[lp[e]] : (isC[e] ? Aub[rc[e]] : 0.0);
        type[e] = classify_bounds(lo[e], hi[e]);
    }
    // K0(i, j), j < NN: row i of [H ; A]. Addresses are rebuilt from a lane id re-materialised next to the loads (see pmpc_qp_reg.hpp).
    auto lane_near = [](int zo) -> unsigned { unsigned l; asm("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l) : "v"(zo)); return l; };
    constexpr int LDH = STACKED ? N : NN;
    // unstacked inputs (the QP entry point): per-slot row base and stride — ONE load per entry (H for primal rows, A for constraint rows)
    const double* rowp[2]; int rstride[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) { rowp[e] = isP[e] ? (H + idx[e]) : (A + rc[e]); rstride[e] = isP[e] ? NN : (isC[e] ? MM : 0); }
    auto Krow = [&](int j, int e, int zo, bool lower) -> double {   // (H or A)(row i_e, j), j < NN; 0 for rows >= N
        if constexpr (STACKED) {
            const unsigned i = lane_near(zo) + 64u * (unsigned)e;
            const bool live = i < (unsigned)N;
            const unsigned ic = live ? i : 0u;
            unsigned b = (lower && ic < (unsigned)j) ? ((unsigned)j + ic * (unsigned)N) : (ic + (unsigned)(j * N));   // (lower: H(max, min); rows >= NN are never < j < NN)
            b += (unsigned)zo; asm("" : "+v"(b));
            const double v = H[b];
            return live ? v : 0.0;
        } else {
            unsigned so = (unsigned)(rstride[e] + zo); asm volatile("" : "+v"(so));   // (volatile: the products j * stride are formed next to their load, not hoisted and spilled)
            const bool up = lower && isP[e] && idx[e] < j;   // lower: H(max, min) — H(j, row) sits at row * NN + j; one load from a selected offset
            const double* base = up ? H : rowp[e];
            const size_t off = up ? ((size_t)(unsigned)lp[e] * NN + (size_t)(unsigned)(j + zo)) : (size_t)((unsigned)j * so);
            const double v = base[off];
            return (isP[e] || isC[e]) ? v : 0.0;
        }
    };
    auto Acol = [&](int k, int e, int zo) -> double {   // A(k, i_e) on primal rows, 0 elsewhere
        if constexpr (STACKED) {
            const unsigned i = lane_near(zo) + 64u * (unsigned)e;
            const bool prim = i < (unsigned)NN;
            unsigned b = (prim ? i : 0u) * (unsigned)N + (unsigned)NN + (unsigned)zo; asm("" : "+v"(b));
            const double v = H[b + (unsigned)k];
            return prim ? v : 0.0;
        } else {
            const double v = A[(size_t)lp[e] * MM + (unsigned)(k + zo)];
            return isP[e] ? v : 0.0;
        }
    };
    constexpr bool XT = (NN > 64) && (NN - 64 <= 4) && (NN - 64 + MM <= 64);   // see the residual evaluation: few primal rows in the second slot, every constraint row there too
    auto AcolAt = [&](int k, int col, int zo) -> double {   // A(k, col) for a per-lane k and a uniform column (col < NN)
        if constexpr (STACKED) { unsigned b = (unsigned)col * (unsigned)N + (unsigned)NN + (unsigned)k + (unsigned)zo; asm("" : "+v"(b)); return H[b]; }
        else return A[(size_t)col * MM + (unsigned)(k + zo)];
    };
    auto xbc = [&](const double (&v)[2], int j) -> double { return (j < 64) ? bcast_lane(v[0], j & 63) : bcast_lane(v[1], (j - 64) & 63); };   // entry j of a two-slot vector

    // state: xv = x (primal rows) / z (constraint rows); yv = y_box / y_a; qv = q (primal rows)
    double xv[2], yv[2], qv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const double x0v = x0 ? x0[lp[e]] : 0.0, ybv = y0 ? y0[MM + lp[e]] : 0.0, yav = y0 ? y0[rc[e]] : 0.0;
        xv[e] = isP[e] ? x0v : 0.0; qv[e] = xv[e]; yv[e] = isP[e] ? ybv : (isC[e] ? yav : 0.0);
    }
    if (x0) {  // z = A * x_guess
        double acc[2] = {0.0, 0.0};
        for (int j = 0; j < NN; ++j) {
            const double xj = (j < 64) ? bcast_uniform(xv[0], j) : bcast_uniform(xv[1], j - 64);
#pragma unroll
            for (int e = 0; e < 2; ++e) acc[e] += Krow(j, e, 0, false) * xj;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) xv[e] = isC[e] ? acc[e] : xv[e];
    }

    double rho = s.rho;
    int rho_updates = 1;
    double rhov[2], rhoinv[2], kdiag[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        rhov[e] = rho_of(type[e], rho);
        rhoinv[e] = 1.0 / rhov[e];
        double kd = H[(size_t)lp[e] * LDH + lp[e]]; kd += s.sigma; kd += rhov[e];
        kdiag[e] = isP[e] ? kd : (isC[e] ? -rhoinv[e] : 0.0);   // (rows >= N: padding, exact zeros, never swept)
    }

    RegKkt2<N> K;
    int status = PMPC_QP_UNSOLVED;
    const double alpha = s.alpha;
    double max_Ax_z_norm = 0.0, max_Hx_ATy_h_norm = 0.0, res_prim = 1.0, res_dual = 1.0, rho_estimate = 0.0;
    int iter = 1;
    int until_check = s.check_termination, until_adapt = s.adaptive_rho_interval;
    bool running = true;
    constexpr int AT_END = 16 * ((NN - 1) / 16 + 1);   // A' entries are consumed in the tile columns that still hold primal rows
    while (running) {
        {   // construct_kkt_matrix + factorise_kkt_matrix (box_admm.hpp:209-223, :336-341)
            const long long f0 = dbg ? clock64() : 0;
            K.invert(ln, tr, kdiag[0], kdiag[1], [&](int j, int e, int z) -> double {
                if (j < NN) return Krow(j < NN ? j : 0, e, z, SYMLOWER);
                if (j >= AT_END) return 0.0;
                return Acol(j >= NN ? j - NN : 0, e, z);
            }, tm);
            if (dbg) dbg[0] += clock64() - f0;
        }
        bool refactor = false;
        // Iterations run in groups that end at the next residual evaluation (iter % check_termination == 0 or iter % adaptive_rho_interval
        // == 0, box_admm.hpp:141 / :160): the inner loop is the mat-vec and the vector updates only, so that the register allocator keeps
        // the whole operand of the mat-vec resident there and confines its spills to the (cold) residual code between the groups.
        while (iter <= s.max_iter) {
            int nrun = s.max_iter - iter + 1;
            if (s.check_termination != 0 && until_check < nrun) nrun = until_check;
            if (s.adaptive_rho && until_adapt < nrun) nrun = until_adapt;
            for (int kk = 0; kk < nrun; ++kk) {
                double rhs[2], zprev[2], sol[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    zprev[e] = xv[e];
                    const double rhsP = ((s.sigma * xv[e] - hv[e]) + rhov[e] * qv[e]) - yv[e];
                    const double rhsC = xv[e] - rhoinv[e] * yv[e];
                    rhs[e] = isP[e] ? rhsP : (isC[e] ? rhsC : 0.0);
                }
                K.apply(rhs[0], rhs[1], tr, ln, sol[0], sol[1]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double zt = zprev[e] + rhoinv[e] * (sol[e] - yv[e]);
                    double zz = alpha * zt;
                    zz += (1 - alpha) * zprev[e] + rhoinv[e] * yv[e];
                    zz = fmin(fmax(zz, lo[e]), hi[e]);
                    const double yC = yv[e] + rhov[e] * ((alpha * zt + (1 - alpha) * zprev[e]) - zz);
                    double xx = alpha * sol[e];
                    xx += (1 - alpha) * xx;  // quirk Q1
                    double qq = xx + rhoinv[e] * yv[e];
                    qq = fmin(fmax(qq, lo[e]), hi[e]);
                    const double yP = yv[e] + rhov[e] * (xx - qq);
                    xv[e] = isP[e] ? xx : (isC[e] ? zz : xv[e]);
                    qv[e] = isP[e] ? qq : qv[e];
                    yv[e] = isP[e] ? yP : (isC[e] ? yC : yv[e]);
                }
            }
            iter += nrun - 1;   // the last iteration performed
            bool check = false, adapt = false;
            if (s.check_termination != 0) { until_check -= nrun; if (until_check == 0) { check = true; until_check = s.check_termination; } }
            if (s.adaptive_rho) { until_adapt -= nrun; if (until_adapt == 0) { adapt = true; until_adapt = s.adaptive_rho_interval; } }
            if (check || adapt) {  // residuals_update, box_admm.hpp:398-415: one add chain per row, columns ascending
                const long long r0 = dbg ? clock64() : 0;
#ifndef PMPC_REG2_RC
#define PMPC_REG2_RC 22   /* measured on config B: 11 / 17 / 22 / 24 / 33 / 44 loads per batch -> 47.9 / 45.4 / 41.0 / 43.5 / 42.9 / 51.0 ms */
#endif
                constexpr int RC = PMPC_REG2_RC;
                int zr = 0;
                asm volatile("" : "+v"(zr));
                double acc[2] = {0.0, 0.0}, aty[2] = {0.0, 0.0};
                bool sparse = false;
                if constexpr (HASJ) {   // (a non-finite iterate takes the dense loops: 0 * inf = NaN on the structural zeros of A)
                    const double probe = ((xv[0] - xv[0]) + (yv[0] - yv[0])) + ((xv[1] - xv[1]) + (yv[1] - yv[1]));
                    sparse = __builtin_amdgcn_ballot_w64(probe != 0.0) == 0;
                }
                constexpr int PUSED = NN + MM + ((NN > 64 && NN - 64 <= 4) ? (NN - 64) * NN : 0);   // x, y, products of the few primal rows of the second slot
                constexpr int NLPK = RegKkt2<N>::lds_parked(PUSED);
                double parked[(RegKkt2<N>::NV - NLPK) * 4 > 0 ? (RegKkt2<N>::NV - NLPK) * 4 : 1];
                K.template park<NLPK, PUSED>(parked, zr, tr + RegKkt2<N>::RHS_OFF, lane_near(zr));
                sched_fence();
                if (HASJ && sparse) {
                  if constexpr (HASJ) {
                    // H x from the workspace (dense), A x and A' y from the block-sparse view in LDS — the same products in the same order.
                    constexpr int NP1 = NN > 64 ? NN - 64 : 0;          // primal rows of the second slot
                    constexpr bool FEW1 = NP1 <= 4;                     // few of them: products through LDS; otherwise their lanes load their rows
                    double* xs = tr + RegKkt2<N>::RHS_OFF; double* ys = xs + NN; double* pb = ys + MM;   // (the staging is free between factorisations)
                    static_assert(RegKkt2<N>::RHS_OFF + NN + MM + (FEW1 ? NP1 * NN : 0) <= RegKkt2<N>::TRI, "residual scratch fits the staging");
                    // the lane's rows and their roles, re-derived from a lane id that is materialised HERE: the long-lived copies (idx, isP, rc, ...) are
                    // spilled across the ADMM loop, and every scratch reload in this block would be an exposed memory round trip
                    const int sl = (int)lane_near(zr);
                    int sidx[2], src[2]; bool sP[2], sC[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) { sidx[e] = sl + 64 * e; sP[e] = sidx[e] < NN; sC[e] = sidx[e] >= NN && sidx[e] < N; src[e] = sC[e] ? sidx[e] - NN : 0; }
#pragma unroll
                    for (int e = 0; e < 2; ++e) { if (sP[e]) xs[sidx[e]] = xv[e]; if (sC[e]) ys[src[e]] = yv[e]; }
                    lds_order();
                    // A' y on the primal rows, A x on the constraint rows — from LDS, before the loads of H occupy the registers
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const bool slotP = (e == 0) || NN > 64, slotC = (e == 0) ? (NN < 64) : true;
                        if (slotP) { const double v = jv.coldot(sP[e] ? sidx[e] : 0, ys, sP[e]); aty[e] = sP[e] ? v : 0.0; }
                        if (slotC) { const double v = jv.rowdot(src[e], xs); acc[e] = sC[e] ? v : 0.0; }
                        sched_fence();
                    }
                    // rows 64 .. NN-1 of H: lane j loads H(64 + t, j) (and lane j < NP1 also H(64 + t, 64 + j)) and forms the product with its own x_j;
                    // lane t then adds the NN products of row 64 + t in ascending j — instead of NN loads per lane for NP1 live lanes
                    double h0[NP1 > 0 ? NP1 : 1], h1[NP1 > 0 ? NP1 : 1];
#pragma unroll
                    for (int t = 0; t < (FEW1 ? NP1 : 0); ++t) {
                        const unsigned l = lane_near(zr);
                        unsigned b0 = (64u + (unsigned)t) + l * (unsigned)N + (unsigned)zr; asm("" : "+v"(b0));
                        h0[t] = H[b0];
                        unsigned b1 = (64u + (unsigned)t) + (64u + (l < (unsigned)NP1 ? l : 0u)) * (unsigned)N + (unsigned)zr; asm("" : "+v"(b1));
                        h1[t] = H[b1];
                    }
                    // rows of the first slot: RCS loads in flight per batch (primal rows only; a constraint row of this slot re-reads row 0)
#ifndef PMPC_REG2_RCS
#define PMPC_REG2_RCS 22
#endif
                    constexpr int RCS = PMPC_REG2_RCS;
                    double hx = 0.0;
#pragma unroll
                    for (int j0 = 0; j0 < NN; j0 += RCS) {
                        double mm[RCS];
#pragma unroll
                        for (int j = 0; j < RCS; ++j) {
                            const unsigned l = lane_near(zr);
                            unsigned b = (l < (unsigned)NN ? l : 0u) + (unsigned)(((j0 + j < NN) ? j0 + j : 0) * N) + (unsigned)zr; asm("" : "+v"(b));
                            mm[j] = H[b];
                        }
#pragma unroll
                        for (int j = 0; j < RCS; ++j) if (j0 + j < NN) hx += mm[j] * xbc(xv, j0 + j);
                        sched_fence();
                    }
                    acc[0] = sP[0] ? hx : acc[0];
                    if constexpr (NP1 > 0 && !FEW1) {   // many primal rows in the second slot: lane l < NP1 loads row 64 + l (the other lanes re-read row 64)
                        double hx1 = 0.0;
#pragma unroll
                        for (int j0 = 0; j0 < NN; j0 += RCS) {
                            double mm[RCS];
#pragma unroll
                            for (int j = 0; j < RCS; ++j) {
                                const unsigned l = lane_near(zr);
                                unsigned b = 64u + (l < (unsigned)NP1 ? l : 0u) + (unsigned)(((j0 + j < NN) ? j0 + j : 0) * N) + (unsigned)zr; asm("" : "+v"(b));
                                mm[j] = H[b];
                            }
#pragma unroll
                            for (int j = 0; j < RCS; ++j) if (j0 + j < NN) hx1 += mm[j] * xbc(xv, j0 + j);
                            sched_fence();
                        }
                        acc[1] = (sl < NP1) ? hx1 : acc[1];
                    }
                    if constexpr (NP1 > 0 && FEW1) {
#pragma unroll
                        for (int t = 0; t < NP1; ++t) { pb[t * NN + sl] = h0[t] * xv[0]; if (sl < NP1) pb[t * NN + 64 + sl] = h1[t] * xv[1]; }
                        lds_order();
                        const int tt = sl < NP1 ? sl : 0;
                        double sacc = 0.0;
#pragma unroll
                        for (int j0 = 0; j0 < NN; j0 += 36) {   // (the tiles are parked: registers for 36 reads in flight)
                            double pv[36];
#pragma unroll
                            for (int j = 0; j < 36; ++j) pv[j] = pb[tt * NN + ((j0 + j < NN) ? j0 + j : 0)];
#pragma unroll
                            for (int j = 0; j < 36; ++j) if (j0 + j < NN) sacc += pv[j];
                        }
                        acc[1] = (sl < NP1) ? sacc : acc[1];
                    }
                    lds_order();
                  }
                } else {
                // one slot at a time, RC loads in flight (the VGPR-resident part of the mat-vec operand is parked in scratch meanwhile)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
#pragma unroll
                    for (int j0 = 0; j0 < NN; j0 += RC) {
                        double mm[RC];
#pragma unroll
                        for (int j = 0; j < RC; ++j) mm[j] = Krow((j0 + j < NN) ? j0 + j : 0, e, zr, false);
#pragma unroll
                        for (int j = 0; j < RC; ++j) if (j0 + j < NN) acc[e] += mm[j] * xbc(xv, j0 + j);
                        sched_fence();
                    }
                    if (e == 1 && NN > 64 && XT) {
                        // The few primal rows of the second slot (config B: rows 64 and 65) need sum_k A(k, row) y_k too. A batch of MM loads per lane for
                        // two live lanes cost two memory round trips per check: instead the lane that OWNS y_k (constraint row k sits on lane NN - 64 + k of
                        // this slot) loads A(k, row) — one coalesced load per row — and forms the product; the products are then added in ascending k on
                        // every lane (v_readlane broadcasts): the same products in the same order.
#pragma unroll
                        for (int t = 0; t < NN - 64; ++t) {
                            const int kk = (int)lane_near(zr) - (NN - 64);
                            const bool own = kk >= 0 && kk < MM;
                            const double av = AcolAt(own ? kk : 0, 64 + t, zr);
                            const double prod = av * yv[1];
                            double sacc = 0.0;
#pragma unroll
                            for (int k = 0; k < MM; ++k) sacc += bcast_lane(prod, NN - 64 + k);
                            aty[1] = ((int)lane_near(zr) == t) ? sacc : aty[1];
                        }
                    } else if (e == 0 || NN > 64) {
#pragma unroll
                        for (int k0 = 0; k0 < MM; k0 += RC) {
                            double mm[RC];
#pragma unroll
                            for (int k = 0; k < RC; ++k) mm[k] = Acol((k0 + k < MM) ? k0 + k : 0, e, zr);
#pragma unroll
                            for (int k = 0; k < RC; ++k) if (k0 + k < MM) aty[e] += mm[k] * xbc(yv, NN + k0 + k);
                            sched_fence();
                        }
                    }
                }
                }
                sched_fence();
                {   // a second opaque zero: with the one of park() the compiler keeps the 36 slot addresses it formed there alive (in AGPRs and in
                    // scratch) and reloads them one by one in front of every load here — 36 dependent memory round trips
                    int zu = 0;
                    asm volatile("" : "+v"(zu));
                    K.template unpark<NLPK, PUSED>(parked, zu, tr + RegKkt2<N>::RHS_OFF, lane_near(zu));
                }
                double a1 = 0.0, a2 = 0.0, rp = 0.0, rq = 0.0, rd = 0.0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double ax = fabs(xv[e]);
                    a1 = fmax(a1, isC[e] ? fmax(fabs(acc[e]), ax) : (isP[e] ? ax : 0.0));
                    a2 = fmax(a2, isP[e] ? fmax(fmax(fabs(acc[e]), fabs(aty[e])), fmax(fabs(hv[e]), fabs(yv[e]))) : 0.0);
                    rp = fmax(rp, isC[e] ? fabs(acc[e] - xv[e]) : 0.0);
                    rq = fmax(rq, isP[e] ? fabs(xv[e] - qv[e]) : 0.0);
                    rd = fmax(rd, isP[e] ? fabs(((acc[e] + hv[e]) + aty[e]) + yv[e]) : 0.0);
                }
                max_Ax_z_norm = wave_max(a1);
                max_Hx_ATy_h_norm = wave_max(a2);
                res_prim = wave_max(rp) + wave_max(rq);
                res_dual = wave_max(rd);
                sched_fence();
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
                    rho = new_rho;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const double prev = rhov[e];
                        rhov[e] = rho_of(type[e], rho);
                        rhoinv[e] = 1.0 / rhov[e];
                        kdiag[e] = isP[e] ? (kdiag[e] + (rhov[e] - prev)) : (isC[e] ? -rhoinv[e] : 0.0);   // update_kkt_rho, box_admm.hpp:448-452
                    }
                    ++rho_updates;
                    refactor = true;
                    ++iter;
                    break;
                }
            }
            ++iter;
        }
        if (!refactor) running = false;
    }
    if (iter > s.max_iter) status = PMPC_QP_MAX_ITER_EXCEEDED;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (isP[e]) { out_x[idx[e]] = xv[e]; out_y[MM + idx[e]] = yv[e]; }
        if (isC[e]) out_y[rc[e]] = yv[e];
    }
    const bool bad = __builtin_amdgcn_ballot_w64((((xv[0] - xv[0]) + (yv[0] - yv[0])) + ((xv[1] - xv[1]) + (yv[1] - yv[1]))) != 0.0) != 0;   // non-finite x or y
    info.status = status; info.iter = iter; info.rho_updates = rho_updates; info.flags = bad ? PMPC_FLAG_NONFINITE : 0;
    info.rho_estimate = rho_estimate; info.res_prim = res_prim; info.res_dual = res_dual;
}

}  // namespace pmpc
