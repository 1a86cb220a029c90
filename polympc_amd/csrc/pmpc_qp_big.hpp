// polympc_amd — blocked LDL^T for KKT systems that fit neither registers nor LDS (config C: n+m = 464), one wavefront per QP.
//
// Replaces, for the large-instance mode of the fused SQP kernel, the linear algebra behind boxADMM::solve_impl: construct_kkt_matrix
// (box_admm.hpp:209-223), factorise_kkt_matrix (:336-341, Eigen::LDLT) and linear_solver.solve (:123). Same arithmetic as the
// static-order right-looking LDL^T of pmpc_qp.hpp — every entry receives  a_ij <- fma(-c_ik, l_jk, a_ij)  for k ascending with the
// UNSCALED column entry c_ik and the scaled l_jk = c_jk / d_k, the substitutions are the column-oriented fma chains, pivot order 0..N-1 —
// so the CPU restatement is the same PIVOT_STATIC and the results are bit-identical to the unblocked kernels. What changes is the
// schedule and the data layout:
//   * the factor lives in HBM as 16 x 16 tiles of the lower block triangle, tile (I, J) at I(I+1)/2 + J, in TWO copies: row-major tiles
//     (Lr: a lane reads ITS ROW of a tile as 128 contiguous bytes — forward substitution, panel factorisation; an accumulator tile is four
//     coalesced 512-byte loads) and k-major tiles (Lc = the transposes: a lane reads ITS COLUMN contiguously — backward substitution; a
//     tile is directly the B operand of the trailing update). The unblocked kernel streamed the packed trailing triangle once per PIVOT
//     (270 MB per factorisation at 464 rows); here a trailing tile is read and written once per 16 pivots (27 MB).
//   * block column k: the diagonal tile is factorised by 16 lanes (pivot values broadcast with v_readlane), every row below applies the
//     16 pivots to its own 16 entries independently (one lane per row, the diagonal tile's d and l through LDS), then every trailing tile
//     gets ONE rank-16 update on the matrix cores: four v_mfma_f64_16x16x4_f64 (a k-ascending fma chain per entry — verified on gfx950,
//     tests/experiments/mfma_f64_probe.hip — which is exactly the order above), A operand = the negated unscaled panel (-C, kept k-major
//     in a 16 x N scratch strip), B operand = the k-major tile of L.
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
    // per-instance HBM workspace (doubles): [Lr tiles | Lc tiles | -C strip (16 x Npad, k-major)]
    __host__ __device__ static size_t doubles(int N) { return 2 * (size_t)ntiles(N) * 256 + (size_t)TB * nblk(N) * TB; }
    static constexpr int LDS_DOUBLES = 256 + 16;                    // diagonal tile (d on the diagonal, l below) + 16 broadcast slots
};

using big_d4 = double __attribute__((ext_vector_type(4)));

// K (lower block triangle, row-major tiles in W) <- [H + diag ; A, diag]; rows / columns >= N: identity padding
__device__ __forceinline__ void big_build(double* __restrict__ W, int n, int m, const double* __restrict__ H, int ldh, const double* __restrict__ A,
                                          int lda, const double* kdiag) {
    const int ln = lane_id();
    const int N = n + m, nb = BigKkt::nblk(N);
    const int r = ln & 15, cg = ln >> 4;
    for (int I = 0; I < nb; ++I)
        for (int J = 0; J <= I; ++J) {
            double* __restrict__ t = W + (size_t)BigKkt::tidx(I, J) * 256;
            const int i = 16 * I + r;
            double e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * J + 4 * q + cg;
                double v = 0.0;
                if (i < N && j < n && i != j) v = (i < n) ? H[(size_t)j * ldh + i] : A[(size_t)j * lda + (i - n)];
                if (i == j) v = (i < N) ? kdiag[i] : 1.0;
                e[q] = v;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) t[r * 16 + 4 * q + cg] = e[q];
        }
    wfence();
    wsync();
}

// in-place blocked LDL^T of the tiles in W (see the header). dl: BigKkt::LDS_DOUBLES doubles of LDS.
__device__ __forceinline__ void big_factor(double* __restrict__ W, int N, double* dl) {
    const int ln = lane_id();
    const int nb = BigKkt::nblk(N), NPAD = nb * 16;
    const size_t nt = (size_t)BigKkt::ntiles(N);
    double* __restrict__ Lr = W;
    double* __restrict__ Lc = W + nt * 256;
    double* __restrict__ Cn = W + 2 * nt * 256;      // -C strip: entry (t, row) at t * NPAD + row
    const int lr = ln >> 4, lc = ln & 15;
    for (int k = 0; k < nb; ++k) {
        // ---- (a) diagonal tile: right-looking LDL^T on 16 lanes (lane r = row r of the tile)
        {
            double* __restrict__ td = Lr + (size_t)BigKkt::tidx(k, k) * 256;
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
                double* __restrict__ tc = Lc + (size_t)BigKkt::tidx(k, k) * 256;
#pragma unroll
                for (int c = 0; c < 16; ++c) { td[r * 16 + c] = a[c]; tc[c * 16 + r] = a[c]; dl[r * 16 + c] = a[c]; }
            }
            wsync();
        }
        if (k == nb - 1) break;
        // ---- (b) rows below the diagonal tile: 16 pivots applied to the row's own 16 entries (one lane per row)
        for (int row0 = 16 * (k + 1); row0 < NPAD; row0 += WAVE) {
            const int row = row0 + ln;
            const bool live = row < NPAD;
            const int rw = live ? row : row0;
            const int I = rw >> 4, rr = rw & 15;
            double* __restrict__ tr_ = Lr + (size_t)BigKkt::tidx(I, k) * 256 + rr * 16;
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
                double* __restrict__ tc = Lc + (size_t)BigKkt::tidx(I, k) * 256 + rr;
#pragma unroll
                for (int c = 0; c < 16; ++c) { tr_[c] = a[c]; tc[c * 16] = a[c]; Cn[(size_t)c * NPAD + row] = cneg[c]; }
            }
        }
        wfence();
        wsync();
        // ---- (c) trailing tiles (I, J), k < J <= I: T += (-C_I) * L_J^T, one rank-16 update on the matrix cores per tile. Tile rows in
        // groups of four (their A operands stay in registers while J runs), the B operand of a J is loaded once per group.
        for (int I0 = k + 1; I0 < nb; I0 += 4) {
            double av[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int I = (I0 + g < nb) ? I0 + g : nb - 1;
#pragma unroll
                for (int s = 0; s < 4; ++s) av[g][s] = Cn[(size_t)(4 * s + lr) * NPAD + 16 * I + lc];
            }
            const int Jend = (I0 + 3 < nb) ? I0 + 3 : nb - 1;
            for (int J = k + 1; J <= Jend; ++J) {
                const double* __restrict__ tb = Lc + (size_t)BigKkt::tidx(J, k) * 256;
                double bv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) bv[s] = tb[64 * s + ln];
                big_d4 T[4];
                // (tiles of the group with I < J do not exist: their loads are redirected to the group's last tile and the result dropped)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int I = I0 + g;
                    const bool ex = I < nb && I >= J;
                    const double* __restrict__ tt = Lr + (size_t)BigKkt::tidx(ex ? I : Jend, ex ? J : k + 1) * 256;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) T[g][rg] = tt[64 * rg + ln];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int s = 0; s < 4; ++s) T[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g][s], bv[s], T[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int I = I0 + g;
                    if (I < nb && I >= J) {
                        double* __restrict__ tt = Lr + (size_t)BigKkt::tidx(I, J) * 256;
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) tt[64 * rg + ln] = T[g][rg];
                    }
                }
            }
        }
        wfence();
        wsync();
    }
}

// v <- K^{-1} v, v in LDS (N entries; padding rows are not touched). bx: 16 doubles of LDS (broadcast slots).
__device__ __forceinline__ void big_solve(const double* __restrict__ W, int N, double* v, double* bx) {
    const int ln = lane_id();
    const int nb = BigKkt::nblk(N), NPAD = nb * 16;
    const size_t nt = (size_t)BigKkt::ntiles(N);
    const double* __restrict__ Lr = W;
    const double* __restrict__ Lc = W + nt * 256;
    constexpr int GS = 4;   // row slots (of 64 rows) whose loads are issued together
    // ---- forward: blocks ascending; inside a block columns ascending
    for (int J = 0; J < nb; ++J) {
        double xj[16];
        {   // finish x_J on 16 lanes (unit-lower triangular solve with the diagonal tile), then broadcast its 16 entries
            const int r = ln & 15;
            const int row = 16 * J + r;
            const double* __restrict__ td = Lr + (size_t)BigKkt::tidx(J, J) * 256 + r * 16;
            double lrow[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) lrow[c] = td[c];
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
                const double* __restrict__ tr_ = Lr + (size_t)BigKkt::tidx(rw >> 4, J) * 256 + (rw & 15) * 16;
#pragma unroll
                for (int c = 0; c < 16; ++c) L[g][c] = tr_[c];
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
    for (int i = ln; i < N; i += WAVE) v[i] = v[i] / Lr[(size_t)BigKkt::tidx(i >> 4, i >> 4) * 256 + (i & 15) * 17];
    wsync();
    // ---- backward: blocks descending; inside a block columns descending
    for (int J = nb - 1; J >= 0; --J) {
        double xj[16];
        {
            const int r = ln & 15;
            const int row = 16 * J + r;
            const double* __restrict__ tc = Lc + (size_t)BigKkt::tidx(J, J) * 256 + r * 16;   // column r of the diagonal tile: L(16J + c, 16J + r), c = 0..15
            double lcol[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) lcol[c] = tc[c];
            double xr = (row < N) ? v[row] : 0.0;
#pragma unroll
            for (int c = 15; c > 0; --c) {
                const double xc = bcast_lane(xr, c);
                const double up = fma(-lcol[c], xc, xr);
                xr = (r < c) ? up : xr;
            }
            if (ln < 16 && row < N) v[row] = xr;
#pragma unroll
            for (int c = 0; c < 16; ++c) xj[c] = bcast_lane(xr, c);
        }
        for (int row0 = 0; row0 < 16 * J; row0 += GS * WAVE) {
            double L[GS][16], vi[GS];
#pragma unroll
            for (int g = 0; g < GS; ++g) {
                const int row = row0 + g * WAVE + ln;
                const int rw = (row < 16 * J) ? row : 0;
                const double* __restrict__ tc = Lc + (size_t)BigKkt::tidx(J, rw >> 4) * 256 + (rw & 15) * 16;   // L(16J + c, rw), c = 0..15
#pragma unroll
                for (int c = 0; c < 16; ++c) L[g][c] = tc[c];
                vi[g] = v[rw];
            }
#pragma unroll
            for (int g = 0; g < GS; ++g) {
                const int row = row0 + g * WAVE + ln;
#pragma unroll
                for (int c = 15; c >= 0; --c) vi[g] = fma(-L[g][c], xj[c], vi[g]);
                if (row < 16 * J) v[row] = vi[g];
            }
        }
        wsync();
    }
    (void)bx;
}

}  // namespace pmpc
