// polympc_amd — blocked LDL^T for KKT systems that fit neither registers nor LDS (config C: n+m = 464), one wavefront per QP.
//
// Replaces, for the large-instance mode of the fused SQP kernel, the linear algebra behind boxADMM::solve_impl: construct_kkt_matrix
// (box_admm.hpp:209-223), factorise_kkt_matrix (:336-341, Eigen::LDLT) and linear_solver.solve (:123). Same arithmetic as the
// static-order right-looking LDL^T of pmpc_qp.hpp — every entry receives  a_ij <- fma(-c_ik, l_jk, a_ij)  for k ascending with the
// UNSCALED column entry c_ik and the scaled l_jk = c_jk / d_k, the substitutions are the column-oriented fma chains, pivot order 0..N-1 —
// and the CPU restatement (PIVOT_BLOCKED) shares PIVOT_STATIC's factorisation and forward pass. What changes is the schedule and the data layout:
//   * the WORKING matrix lives in HBM as 16 x 16 row-major tiles of the lower block triangle, tile (I, J) at I(I+1)/2 + J (Lr: an MFMA
//     accumulator tile is four coalesced 512-byte loads). The unblocked kernel streamed the packed trailing triangle once per PIVOT
//     (270 MB per factorisation at 464 rows), a right-looking tile schedule once per 16 pivots (27 MB read and written); the left-looking
//     schedule used here reads and writes every tile once and streams the operands of its updates instead (see big_factor).
//   * the finished FACTOR is written once, as column panels LF — per block column J the 16 columns of L below (and including) the diagonal tile,
//     each column contiguous over the rows, in slabs of 64 rows: with one lane per row every load instruction is a contiguous 512-byte segment.
//     The forward substitution streams it (instruction c loads L(row, 16J + c) for 64 consecutive rows), it is the B operand of the trailing
//     update, and the BACKWARD substitution streams the same panels: the contributions of the rows below a block are column dot products
//     (per-lane partial sums, combined in a fixed order) instead of the row-oriented fma chains, which need a second, row-ordered copy of L and
//     twice the traffic per ADMM iteration (measured with that copy: 195 GB per launch, 99 GB of it in the substitutions). The summation order of
//     the backward pass is therefore its own restated policy, PIVOT_BLOCKED; factor and forward pass are PIVOT_STATIC's operation for operation.
//   * block column k: the diagonal tile is factorised by 16 lanes (pivot values broadcast with v_readlane), every row below applies the
//     16 pivots to its own 16 entries independently (one lane per row, the diagonal tile's d and l through LDS), then every trailing tile
//     gets ONE rank-16 update on the matrix cores: four v_mfma_f64_16x16x4_f64 (a k-ascending fma chain per entry — verified on gfx950,
//     tests/experiments/mfma_f64_probe.hip — which is exactly the order above), A operand = the negated unscaled panel (-C, kept k-major
//     in the CF strip of its block column), B operand = the k-major tile of L.
//   * substitutions: lane per row; per block column the 16 finished entries are broadcast (v_readlane) and every row below (above) applies
//     its 16 fma from ONE contiguous 128-byte load; 32 such loads are in flight per batch. 2N dependent steps become 2N/16.
// MFMA-busy is what bounds a single wavefront here (33 MFLOP per factorisation at 32 flop/cycle/SIMD), HBM traffic what bounds the batch
// (factor + two substitution passes per ADMM iteration: 1.7 MB per iteration and instance).
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_qp.hpp"
#include "pmpc_qp_reg.hpp"

namespace pmpc {

struct BigKkt {
    static constexpr int TB = 16;                                   // tile edge
    __host__ __device__ static int nblk(int N) { return (N + TB - 1) / TB; }
    __host__ __device__ static int ntiles(int N) { const int nb = nblk(N); return nb * (nb + 1) / 2; }
    __host__ __device__ static int tidx(int I, int J) { return I * (I + 1) / 2 + J; }
    // Panels are stored in SLABS of 64 lanes x 16 entries (8 KB): entry (c, rel) of a panel at (rel / 64) * 1024 + c * 64 + rel % 64, so that the 16
    // load instructions of one lane-per-row slot sweep ONE contiguous 8 KB region (sixteen 512-byte pieces a panel-column apart kept one DRAM
    // row per piece open).
    //   LF, block column J: rel = row - 16J, c = column - 16J, (NPAD - 16J) rows padded to a multiple of 64; panels in J order
    __host__ __device__ static size_t sizeF(int J, int NPAD) { return (size_t)16 * (((NPAD - 16 * J) + 63) / 64 * 64); }
    __host__ __device__ static size_t sizeB(int J) { return (size_t)16 * ((16 * (J + 1) + 63) / 64 * 64); }
    __host__ __device__ static size_t ceil4_sum(int t) { const int Q = t >> 2, R = t & 3; return (size_t)(Q + 1) * (2 * Q + R); }   // sum_{u=1..t} ceil(u / 4)
    __host__ __device__ static size_t offB(int J) { return 1024 * ceil4_sum(J); }                                    // sizeB(j) = 1024 ceil((j+1)/4)
    __host__ __device__ static size_t offF(int J, int NPAD) { const int nb = NPAD >> 4; return 1024 * (ceil4_sum(nb) - ceil4_sum(nb - J)); }   // sizeF(j) = 1024 ceil((nb-j)/4)
    __host__ __device__ static size_t slab(int c, int rel) { return (size_t)(rel >> 6) * 1024 + (size_t)c * 64 + (rel & 63); }
    //   CF, block column k: the NEGATED UNSCALED column entries -c of the rows below the diagonal tile (the A operand of the tile updates), k-major:
    //   entry (t, row) at t * (NPAD - 16(k+1)) + row - 16(k+1); strips in k order
    __host__ __device__ static size_t offC(int k, int NPAD) { return (size_t)16 * ((size_t)k * NPAD - (size_t)8 * k * (k + 1)); }
    // per-instance HBM workspace (doubles): [Lr working tiles | LF column panels | CF strips]
    __host__ __device__ static size_t doubles(int N) {
        const int nb = nblk(N);
        return (size_t)ntiles(N) * 256 + offF(nb, nb * 16) + offC(nb > 0 ? nb - 1 : 0, nb * 16) + 16;
    }
    static constexpr int LDS_DOUBLES = 256 + 16 + 16 * 65 + 64;     // diagonal tile (d on the diagonal, l below) + 16 slots + the backward pass's partial sums (16 columns, stride 65: the group sums read 16 columns at once) and group sums (4 x 16)
};

using big_d4 = double __attribute__((ext_vector_type(4)));

// K (lower block triangle, row-major tiles in W) <- [H + diag ; A, diag]; rows / columns >= N: identity padding.
// Eight tiles of a tile row per pass (32 independent loads in flight): one tile at a time was a chain of 435 dependent load -> store round trips at 464 rows
// (2.4 M cycles per factorisation, 8.5 % of config C).
__device__ __forceinline__ void big_build(double* W, int n, int m, const double* __restrict__ H, int ldh, const double* __restrict__ A,
                                          int lda, const double* kdiag) {
    const int ln = lane_id();
    const int N = n + m, nb = BigKkt::nblk(N);
    const int r = ln & 15, cg = ln >> 4;
    constexpr int G = 8;
    for (int I = 0; I < nb; ++I) {
        const int i = 16 * I + r;
        const double kd = (i < N) ? kdiag[i] : 1.0;
        for (int J0 = 0; J0 <= I; J0 += G) {
            double e[G][4];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int J = (J0 + g <= I) ? J0 + g : I;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = 16 * J + 4 * q + cg;
                    const bool off = i < N && j < n && i != j;
                    const double* src = (i < n) ? H + (size_t)j * ldh + i : A + (size_t)j * lda + (i - n);
                    const double v = off ? *src : 0.0;
                    e[g][q] = (i == j) ? kd : v;
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int J = J0 + g;
                if (J <= I) {
                    double* t = W + (size_t)BigKkt::tidx(I, J) * 256;
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[r * 16 + 4 * q + cg] = e[g][q];
                }
            }
        }
    }
    wfence();
    wsync();
}

// in-place blocked LDL^T of the tiles in W (see the header). dl: BigKkt::LDS_DOUBLES doubles of LDS.
// LEFT-LOOKING schedule: block column J first receives the rank-16 updates of ALL earlier block columns k < J (accumulator tiles stay in
// registers while k runs: per update one A operand tile from the CF strip of k and a quarter of a B operand tile from the LF panel of k are read),
// then its diagonal tile is factorised and the rows below apply the 16 pivots. Every entry still receives fma(-c_ik, l_jk, a_ij) for k ascending
// — the right-looking schedule's operations in the right-looking schedule's order — but a tile is read and written ONCE instead of once per earlier
// block column (config C: 48 GB of the 220 GB a launch moved were those writes).
__device__ __forceinline__ void big_factor(double* W, int N, double* dl) {
    const int ln = lane_id();
    const int nb = BigKkt::nblk(N), NPAD = nb * 16;
    const size_t nt = (size_t)BigKkt::ntiles(N);
    double* Lr = W;
    double* LF = W + nt * 256;
    double* CF = LF + BigKkt::offF(nb, NPAD);
    const int lr = ln >> 4, lc = ln & 15;
    size_t oF = 0;
    for (int J = 0; J < nb; oF += BigKkt::sizeF(J, NPAD), ++J) {
        double* pF = LF + oF;   // forward panel of block column J
        // ---- (u) tiles (I, J), I >= J: T += sum_k (-C_I^k) * (L_J^k)^T, k ascending, four v_mfma_f64_16x16x4_f64 per k. Tile rows in groups of
        // four; the operands of k + 1 are requested before the matrix cores work on k.
        if (J > 0) {
            for (int I0 = J; I0 < nb; I0 += 4) {
                big_d4 T[4];
                int rowI[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int I = (I0 + g < nb) ? I0 + g : nb - 1;   // (a group's missing tiles repeat its last one; their result is dropped)
                    rowI[g] = 16 * I + lc;
                    const double* tt = Lr + (size_t)BigKkt::tidx(I, J) * 256;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) T[g][rg] = tt[64 * rg + ln];
                }
                double av[4][4], bv[4];
                auto load_ops = [&](int k, double (&a)[4][4], double (&b)[4]) {
                    const double* cs = CF + BigKkt::offC(k, NPAD);
                    const int w = NPAD - 16 * (k + 1);
                    const double* pk = LF + BigKkt::offF(k, NPAD);
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) b[sx] = pk[BigKkt::slab(4 * sx + lr, 16 * (J - k) + lc)];   // B(kk = 4s + lr, col = lc) = L(16J + lc, 16k + 4s + lr)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int sx = 0; sx < 4; ++sx) a[g][sx] = cs[(size_t)(4 * sx + lr) * w + rowI[g] - 16 * (k + 1)];
                };
                load_ops(0, av, bv);
                for (int k = 0; k < J; ++k) {
                    double an[4][4], bn[4];
                    load_ops((k + 1 < J) ? k + 1 : k, an, bn);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int sx = 0; sx < 4; ++sx) T[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g][sx], bv[sx], T[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int sx = 0; sx < 4; ++sx) av[g][sx] = an[g][sx];
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) bv[sx] = bn[sx];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int I = I0 + g;
                    if (I < nb) {
                        double* tt = Lr + (size_t)BigKkt::tidx(I, J) * 256;
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) tt[64 * rg + ln] = T[g][rg];
                    }
                }
            }
            wfence();
            wsync();
        }
        // ---- (a) diagonal tile: right-looking LDL^T on 16 lanes (lane r = row r of the tile)
        {
            double* td = Lr + (size_t)BigKkt::tidx(J, J) * 256;
            const int r = ln & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = td[r * 16 + c];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double dt = bcast_lane(a[t], t);
                const double c = a[t];
                const double l = c / dt;
#pragma unroll
                for (int u = t + 1; u < 16; ++u) {
                    const double lut = bcast_lane(l, u);
                    const double upd = fma(-c, lut, a[u]);
                    a[u] = (r >= u) ? upd : a[u];
                }
                a[t] = (r > t) ? l : a[t];
            }
            if (ln < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) { pF[BigKkt::slab(c, r)] = a[c]; dl[r * 16 + c] = a[c]; }
            }
            wfence();
            wsync();
        }
        if (J == nb - 1) break;
        // ---- (b) rows below the diagonal tile: 16 pivots applied to the row's own 16 entries (one lane per row)
        double* cs = CF + BigKkt::offC(J, NPAD);
        const int w = NPAD - 16 * (J + 1);
        for (int row0 = 16 * (J + 1); row0 < NPAD; row0 += WAVE) {
            const int row = row0 + ln;
            const bool live = row < NPAD;
            const int rw = live ? row : row0;
            const int I = rw >> 4, rr = rw & 15;
            double* tr_ = Lr + (size_t)BigKkt::tidx(I, J) * 256 + rr * 16;
            double a[16], cneg[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = tr_[c];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double c = a[t];
                const double l = c / dl[t * 16 + t];
#pragma unroll
                for (int u = t + 1; u < 16; ++u) a[u] = fma(-c, dl[u * 16 + t], a[u]);
                cneg[t] = -c;
                a[t] = l;
            }
            if (live) {
#pragma unroll
                for (int c = 0; c < 16; ++c) { pF[BigKkt::slab(c, row - 16 * J)] = a[c]; cs[(size_t)c * w + row - 16 * (J + 1)] = cneg[c]; }
            }
        }
        wfence();
        wsync();
    }
}

// v <- K^{-1} v, v in LDS (N entries; padding rows are not touched). bx: unused LDS slots.
__device__ __forceinline__ void big_solve(const double* W, int N, double* v, double* bx) {
    const int ln = lane_id();
    const int nb = BigKkt::nblk(N), NPAD = nb * 16;
    const size_t nt = (size_t)BigKkt::ntiles(N);
    const double* LF = W + nt * 256;
#ifndef PMPC_BIG_GS
#define PMPC_BIG_GS 4
#endif
    constexpr int GS = PMPC_BIG_GS;   // row slots (of 64 rows) whose loads are issued together
    // ---- forward: blocks ascending; inside a block columns ascending
    size_t oF = 0;
    for (int J = 0; J < nb; oF += BigKkt::sizeF(J, NPAD), ++J) {
        const double* pF = LF + oF;
        double xj[16];
        {   // finish x_J on 16 lanes (unit-lower triangular solve with the diagonal tile), then broadcast its 16 entries
            const int r = ln & 15;
            const int row = 16 * J + r;
            double lrow[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) lrow[c] = pF[BigKkt::slab(c, r)];
            double xr = (row < N) ? v[row] : 0.0;
#pragma unroll
            for (int c = 0; c < 15; ++c) {
                const double xc = bcast_lane(xr, c);
                const double up = fma(-lrow[c], xc, xr);
                xr = (r > c) ? up : xr;
            }
            if (ln < 16 && row < N) v[row] = xr;
#pragma unroll
            for (int c = 0; c < 16; ++c) xj[c] = bcast_lane(xr, c);
        }
        for (int row0 = 16 * (J + 1); row0 < NPAD; row0 += GS * WAVE) {
            double L[GS][16], vi[GS];
#pragma unroll
            for (int g = 0; g < GS; ++g) {
                const int row = row0 + g * WAVE + ln;
                const int rw = (row < NPAD) ? row : NPAD - 1;
#pragma unroll
                for (int c = 0; c < 16; ++c) L[g][c] = pF[BigKkt::slab(c, rw - 16 * J)];
                vi[g] = (row < N) ? v[row] : 0.0;
            }
#pragma unroll
            for (int g = 0; g < GS; ++g) {
                const int row = row0 + g * WAVE + ln;
#pragma unroll
                for (int c = 0; c < 16; ++c) vi[g] = fma(-L[g][c], xj[c], vi[g]);
                if (row < N) v[row] = vi[g];
            }
        }
        wsync();
    }
    // ---- diagonal
    for (int i = ln; i < N; i += WAVE) {
        const int I = i >> 4, r = i & 15;
        v[i] = v[i] / LF[BigKkt::offF(I, NPAD) + BigKkt::slab(r, r)];
    }
    wsync();
    // ---- backward, from the SAME column panels (no row-ordered copy of L is read): per block of 16 columns, descending, the contributions of
    // the rows below the block are column dot products — every lane keeps one partial sum per column over its rows (row 16(J+1) + 64 g + lane,
    // g ascending, fma); the 64 partials of a column are added in four groups of 16 (lane (c, k) adds group k of column c in index order), the
    // group sums as (S0 + S1) + (S2 + S3) — subtracted once; then the block's own triangle, columns descending. (CPU restatement: PIVOT_BLOCKED.)
    constexpr int GB = 2;     // row slots in flight (16 running sums per lane on top of the loaded entries)
    double* red = bx + 16;    // 16 x 64 partial sums, column stride 65 (conflict-free for the column-parallel group sums)
    double* grp = red + 16 * 65; // 4 x 16 group sums
    size_t oFb = BigKkt::offF(nb, NPAD);
    for (int J = nb - 1; J >= 0; --J) {
        oFb -= BigKkt::sizeF(J, NPAD);
        const double* pF = LF + oFb;
        double acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.0;
        for (int row0 = 16 * (J + 1); row0 < NPAD; row0 += GB * WAVE) {
            double L[GB][16], xr[GB];
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int row = row0 + g * WAVE + ln;
                const int rw = (row < NPAD) ? row : NPAD - 1;
#pragma unroll
                for (int c = 0; c < 16; ++c) L[g][c] = pF[BigKkt::slab(c, rw - 16 * J)];
                xr[g] = (row < N) ? v[row] : 0.0;
            }
#pragma unroll
            for (int g = 0; g < GB; ++g)
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fma(L[g][c], xr[g], acc[c]);   // (rows beyond the matrix carry x = 0: the sum is unchanged)
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) red[c * 65 + ln] = acc[c];
        wsync();
        const int r = ln & 15, kq = ln >> 4;
        {
            double t[16], a = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = red[r * 65 + 16 * kq + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += t[u];
            grp[kq * 16 + r] = a;
        }
        wsync();
        {
            const int row = 16 * J + r;
            const double sum = (grp[r] + grp[16 + r]) + (grp[32 + r] + grp[48 + r]);
            double lcol[16];   // column r of the diagonal tile: L(16J + c, 16J + r), c = 0..15 — contiguous in the column panel
#pragma unroll
            for (int c = 0; c < 16; ++c) lcol[c] = pF[BigKkt::slab(r, c)];
            double xr = (row < N) ? v[row] : 0.0;
            xr = xr - sum;
#pragma unroll
            for (int c = 15; c > 0; --c) {
                const double xc = bcast_lane(xr, c);
                const double up = fma(-lcol[c], xc, xr);
                xr = (r < c) ? up : xr;
            }
            if (ln < 16 && row < N) v[row] = xr;
        }
        wsync();
    }
    (void)bx;
}

}  // namespace pmpc
