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
//     its 16 fma from eight 16-byte loads (column pairs side by side in the panel, see BigKkt::slab); 32 such loads are in flight per batch.
//     2N dependent steps become 2N/16.
// MFMA-busy is what bounds a single wavefront here (33 MFLOP per factorisation at 32 flop/cycle/SIMD), HBM traffic what bounds the batch
// (factor + two substitution passes per ADMM iteration: 1.7 MB per iteration and instance).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "pmpc_qp.hpp"
#include "pmpc_qp_reg.hpp"

namespace pmpc {

struct BigKkt {
    static constexpr int TB = 16;                                   // tile edge
    __host__ __device__ static int nblk(int N) { return (N + TB - 1) / TB; }
    __host__ __device__ static int ntiles(int N) { const int nb = nblk(N); return nb * (nb + 1) / 2; }
    __host__ __device__ static int tidx(int I, int J) { return I * (I + 1) / 2 + J; }
    // Panels are stored in SLABS of 64 lanes x 16 entries (8 KB), so that the load instructions of one lane-per-row slot sweep ONE contiguous 8 KB
    // region (sixteen 512-byte pieces a panel-column apart kept one DRAM row per piece open); entry (c, rel) of a panel: slab(c, rel) below.
    //   LF, block column J: rel = row - 16J, c = column - 16J, (NPAD - 16J) rows padded to a multiple of 64; panels in J order
    __host__ __device__ static size_t sizeF(int J, int NPAD) { return (size_t)16 * (((NPAD - 16 * J) + 63) / 64 * 64); }
    __host__ __device__ static size_t sizeB(int J) { return (size_t)16 * ((16 * (J + 1) + 63) / 64 * 64); }
    __host__ __device__ static size_t ceil4_sum(int t) { const int Q = t >> 2, R = t & 3; return (size_t)(Q + 1) * (2 * Q + R); }   // sum_{u=1..t} ceil(u / 4)
    __host__ __device__ static size_t offB(int J) { return 1024 * ceil4_sum(J); }                                    // sizeB(j) = 1024 ceil((j+1)/4)
    __host__ __device__ static size_t offF(int J, int NPAD) { const int nb = NPAD >> 4; return 1024 * (ceil4_sum(nb) - ceil4_sum(nb - J)); }   // sizeF(j) = 1024 ceil((nb-j)/4)
    // (round 4) inside a slab the 16 columns are stored as 8 PAIRS: entries (2p, rel) and (2p + 1, rel) side by side, so that a lane-per-row slot reads
    // its 16 entries with eight 16-byte loads — a wavefront's load instructions cost ~55 cycles each whatever their width (tests/experiments/
    // wave_stream_probe.hip: 8-byte loads stream 9 B/cycle at any depth, 16-byte loads 20 B/cycle), and the triangular passes are made of them
    __host__ __device__ static size_t slab(int c, int rel) { return (size_t)(rel >> 6) * 1024 + (size_t)(c >> 1) * 128 + (size_t)(rel & 63) * 2 + (c & 1); }
    __host__ __device__ static size_t slab_pair(int p, int rel) { return (size_t)(rel >> 6) * 1024 + (size_t)p * 128 + (size_t)(rel & 63) * 2; }   // 16-byte aligned: columns 2p, 2p + 1
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
using big_d2 = double __attribute__((ext_vector_type(2)));


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

// Condensed mode (round 3): the constraint block of K is diagonal (-1 / rho), so it is eliminated in closed form and the tiles hold the n x n SPD matrix
//     S = H + diag(kdiag[0:n]) + A' diag(rho) A          (x = S^{-1} (r1 + A'(rho o r2)),  nu = rho o (A x - r2): boxadmm_solve)
// instead of the (n + m) x (n + m) KKT matrix — config C: 256 rows instead of 464, a factor of 262 KB instead of 861 KB streamed twice per ADMM
// iteration, a sixth of the factorisation's flops. A' diag(rho) A is a rank-m update on the matrix cores: per block column of S the accumulator tiles
// start from H (+ diagonal) and take, for the constraint rows r in groups of four (ascending — the k-ascending fma chain of v_mfma_f64_16x16x4_f64),
// A operand rho_r J(r, i), B operand J(r, j); the operands are evaluated from the block-sparse view of J (per-node blocks + differentiation
// matrix), the dense J is never read. Restated on the CPU by the test suite as PIVOT_CONDENSED.
// Returns max_i S_ii (the conditioning gate of boxadmm_solve reads it).
template <class JV, int NW>
__device__ __forceinline__ double big_build_condensed(double* W, int n, int m, const double* __restrict__ H, int ldh, const double* kdiag,
                                                      const double* rho, const JV& jv, const BigTeam<NW>& team, double* red4) {
    const int ln = lane_id();
    int item = 0;   // (pair of block columns, group of tile rows): independent pieces, dealt round-robin over the team
    const int nb = BigKkt::nblk(n);
    const int lr = ln >> 4, lc = ln & 15;
    double dmax = 0.0;
    constexpr int GI = 8;    // tile rows per pass (two block columns each: 16 accumulator tiles = 128 registers)
    // TWO block columns per pass (2 x 16 accumulator tiles = 256 registers, the accumulation file): the A operands — the expensive part, an entry
    // lookup per lane and tile row — serve both columns
    for (int Jc = 0; Jc < nb; Jc += 2) {
        const bool two = Jc + 1 < nb;
        const int j0 = 16 * Jc + lc, j1 = 16 * (Jc + 1) + lc;
        const typename JV::Col cj0 = jv.column(j0 < n ? j0 : 0), cj1 = jv.column((two && j1 < n) ? j1 : 0);
        for (int I0 = Jc; I0 < nb; I0 += GI) {
            if (!team.mine(item++)) continue;
            big_d4 T0[GI], T1[GI];
            typename JV::Col ci[GI];
#pragma unroll
            for (int g = 0; g < GI; ++g) {
                const int I = (I0 + g < nb) ? I0 + g : nb - 1;   // (a group's missing tiles repeat its last one; their result is dropped)
                const int ic = 16 * I + lc;
                ci[g] = jv.column(ic < n ? ic : 0);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int i = 16 * I + 4 * rg + lr;
                    const double kd = (i < n) ? kdiag[i < n ? i : 0] : 1.0;
                    const bool in0 = i < n && j0 < n, in1 = two && i < n && j1 < n;
                    const double hv0 = H[(size_t)(in0 ? j0 : 0) * ldh + (in0 ? i : 0)];
                    const double hv1 = H[(size_t)(in1 ? j1 : 0) * ldh + (in1 ? i : 0)];
                    T0[g][rg] = (i == j0) ? kd : (in0 ? hv0 : 0.0);
                    T1[g][rg] = (i == j1) ? kd : (in1 ? hv1 : 0.0);
                }
            }
            for (int r0 = 0; r0 < m; r0 += 4) {
                const int r = r0 + lr;
                const bool rin = r < m;
                const int rc = rin ? r : 0;
                const typename JV::Row rw = jv.rowinfo(rc);
                // a group of four constraint rows without an entry in these block columns leaves every tile of the pass unchanged (fma(a, 0, c) = c):
                // most of them — a row touches the state columns of its own segment and its own node's block only. Decided from the structure
                // alone (no load), so that skipped groups cost a few integer operations
                const bool h0 = rin && j0 < n && jv.structural(rw, cj0), h1 = two && rin && j1 < n && jv.structural(rw, cj1);
                if (__builtin_amdgcn_ballot_w64(h0 || h1) == 0) continue;
                const double b0v = jv.jval(rw, cj0), b1v = jv.jval(rw, cj1);
                const double bop0 = (rin && j0 < n) ? b0v : 0.0, bop1 = (two && rin && j1 < n) ? b1v : 0.0;
                const double rr = rho[rc];
                double aop[GI];
#pragma unroll
                for (int g = 0; g < GI; ++g) {
                    const int I = (I0 + g < nb) ? I0 + g : nb - 1;
                    const int i = 16 * I + lc;
                    const double av = jv.jval(rw, ci[g]);
                    aop[g] = (rin && i < n) ? rr * av : 0.0;
                }
#pragma unroll
                for (int g = 0; g < GI; ++g)
                    if (__builtin_amdgcn_ballot_w64(aop[g] != 0.0) != 0) {
                        T0[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[g], bop0, T0[g], 0, 0, 0);
                        T1[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[g], bop1, T1[g], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int g = 0; g < GI; ++g) {
                const int I = I0 + g;
                if (I < nb) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {   // diagonal entries of S: tile (Jc, Jc) in T0, tile (Jc + 1, Jc + 1) in T1, rows < n
                        const bool dg = (4 * rg + lr == lc) && (16 * I + lc < n);
                        if (I == Jc) dmax = fmax(dmax, dg ? fabs(T0[g][rg]) : 0.0);
                        if (two && I == Jc + 1) dmax = fmax(dmax, dg ? fabs(T1[g][rg]) : 0.0);
                    }
                    double* tt = W + (size_t)BigKkt::tidx(I, Jc) * 256;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) tt[64 * rg + ln] = T0[g][rg];
                    if (two && I >= Jc + 1) {
                        double* t1 = W + (size_t)BigKkt::tidx(I, Jc + 1) * 256;
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) t1[64 * rg + ln] = T1[g][rg];
                    }
                }
            }
        }
    }
    team.sync();
    double dm = wave_max(dmax);
    if constexpr (NW > 1) {
        if (ln == 0) red4[team.w] = dm;
        team.sync();
        dm = fmax(fmax(red4[0], red4[1]), fmax(red4[2], red4[3]));
        team.sync();
    }
    return dm;
}

// in-place blocked LDL^T of the tiles in W (see the header). dl: BigKkt::LDS_DOUBLES doubles of LDS. Returns the smallest |pivot| over the rows < N.
// LEFT-LOOKING schedule, two block columns at a time: block columns J and J + 1 first receive the rank-16 updates of ALL earlier block columns k < J
// together (accumulator tiles of both columns stay in registers while k runs: per k and group of four tile rows, four A operand tiles from the CF
// strip of k and one B operand tile per column from the LF panel of k feed 32 MFMA — 12 KB per 32 MFMA, where one column at a time needs 20 KB and the
// right-looking schedule re-read and re-wrote every tile per k); then column J is finished (diagonal tile, the rows below apply the 16 pivots), column
// J + 1 takes its last update (k = J) and is finished. Every entry still receives fma(-c_ik, l_jk, a_ij) for k ascending — the right-looking
// schedule's operations in the right-looking schedule's order.
template <int NW>
__device__ __forceinline__ double big_factor(double* W, int N, double* dl, const BigTeam<NW>& team) {
    const int ln = lane_id();
    const int nb = BigKkt::nblk(N), NPAD = nb * 16;
    double piv_min = INFINITY;
    const size_t nt = (size_t)BigKkt::ntiles(N);
    double* Lr = W;
    double* LF = W + nt * 256;
    double* CF = LF + BigKkt::offF(nb, NPAD);
    const int lr = ln >> 4, lc = ln & 15;

    // tiles (I, Jc + c), c < NC, I in [I_first, I_end): T += sum_{k in [k0, k1)} (-C_I^k) * (L_{Jc+c}^k)^T, k ascending, four v_mfma_f64_16x16x4_f64 per k and
    // tile. Tile rows in groups of GI; the operands of k + 1 are requested before the matrix cores work on k.
    auto update_cols = [&](auto nc_tag, auto gi_tag, int Jc, int I_first, int I_end, int k0, int k1) {
        constexpr int NC = decltype(nc_tag)::value, GI = decltype(gi_tag)::value;
        if (k1 <= k0) return;
        for (int I0 = I_first; I0 < I_end; I0 += GI) {
            if (!team.mine((I0 - I_first) / GI + Jc)) continue;   // groups of tile rows: independent, dealt round-robin (rotated by the block column)
            big_d4 T[NC][GI];
            int rowI[GI];
#pragma unroll
            for (int g = 0; g < GI; ++g) {
                const int I = (I0 + g < I_end) ? I0 + g : I_end - 1;   // (a group's missing tiles repeat its last one; their result is dropped)
                rowI[g] = 16 * I + lc;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double* tt = Lr + (size_t)BigKkt::tidx(I, Jc + c) * 256;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) T[c][g][rg] = tt[64 * rg + ln];
                }
            }
            double av[GI][4], bv[NC][4];
            auto load_ops = [&](int k, double (&a)[GI][4], double (&b)[NC][4]) {
                const double* cs = CF + BigKkt::offC(k, NPAD);
                const int w = NPAD - 16 * (k + 1);
                const double* pk = LF + BigKkt::offF(k, NPAD);
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) b[c][sx] = pk[BigKkt::slab(4 * sx + lr, 16 * (Jc + c - k) + lc)];   // B(kk = 4s + lr, col = lc) = L(16(Jc+c) + lc, 16k + 4s + lr)
#pragma unroll
                for (int g = 0; g < GI; ++g)
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) a[g][sx] = cs[(size_t)(4 * sx + lr) * w + rowI[g] - 16 * (k + 1)];
            };
            load_ops(k0, av, bv);
            for (int k = k0; k < k1; ++k) {
                double an[GI][4], bn[NC][4];
                load_ops((k + 1 < k1) ? k + 1 : k, an, bn);
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int g = 0; g < GI; ++g)
#pragma unroll
                        for (int sx = 0; sx < 4; ++sx) T[c][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g][sx], bv[c][sx], T[c][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < GI; ++g)
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) av[g][sx] = an[g][sx];
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) bv[c][sx] = bn[c][sx];
            }
#pragma unroll
            for (int g = 0; g < GI; ++g) {
                const int I = I0 + g;
                if (I < I_end) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        double* tt = Lr + (size_t)BigKkt::tidx(I, Jc + c) * 256;
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) tt[64 * rg + ln] = T[c][g][rg];
                    }
                }
            }
        }
    };
    using one = std::integral_constant<int, 1>;
    using two = std::integral_constant<int, 2>;
    using four = std::integral_constant<int, (NW > 1) ? 2 : 4>;   // tile rows per group (a team deals smaller groups: more pieces than wavefronts for longer)
    constexpr int PAIR_MIN_BLOCKS = 16;

    // diagonal tile of block column J, then the rows below it: LF panel J and CF strip J
    auto finish_column = [&](int J) {
        double* pF = LF + BigKkt::offF(J, NPAD);
        // ---- (a) diagonal tile: right-looking LDL^T on 16 lanes (lane r = row r of the tile) — the team's first wavefront
        if (team.lead()) {
            double* td = Lr + (size_t)BigKkt::tidx(J, J) * 256;
            const int r = ln & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = td[r * 16 + c];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double dt = bcast_lane(a[t], t);
                piv_min = (16 * J + t < N) ? fmin(piv_min, fabs(dt)) : piv_min;
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
                for (int c = 0; c < 16; ++c) dl[r * 16 + c] = a[c];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) { big_d2 v2; v2[0] = a[2 * pp]; v2[1] = a[2 * pp + 1]; *reinterpret_cast<big_d2*>(pF + BigKkt::slab_pair(pp, r)) = v2; }
            }
        }
        team.sync();
        if (J == nb - 1) return;
        // ---- (b) rows below the diagonal tile: 16 pivots applied to the row's own 16 entries (one lane per row)
        double* cs = CF + BigKkt::offC(J, NPAD);
        const int w = NPAD - 16 * (J + 1);
        for (int row0 = 16 * (J + 1); row0 < NPAD; row0 += WAVE) {
            if (!team.mine((row0 - 16 * (J + 1)) / WAVE)) continue;   // 64-row slots: independent
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
                for (int c = 0; c < 16; ++c) cs[(size_t)c * w + row - 16 * (J + 1)] = cneg[c];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) { big_d2 v2; v2[0] = a[2 * pp]; v2[1] = a[2 * pp + 1]; *reinterpret_cast<big_d2*>(pF + BigKkt::slab_pair(pp, row - 16 * J)) = v2; }
            }
        }
        team.sync();
    };

    if (nb < PAIR_MIN_BLOCKS) {   // few block columns: one at a time (the extra passes of the paired schedule cost more than the operands they save: 128 rows +4 %)
        for (int J = 0; J < nb; ++J) {
            if (J > 0) { update_cols(one{}, four{}, J, J, nb, 0, J); team.sync(); }
            finish_column(J);
        }
        return piv_min;
    }
    for (int J = 0; J < nb; J += 2) {
        const bool pair = J + 1 < nb;
        if (J > 0) {
            update_cols(one{}, one{}, J, J, J + 1, 0, J);               // tile (J, J): the only tile of row J in this pair (one piece: one wavefront of the team)
            if (pair) update_cols(two{}, four{}, J, J + 1, nb, 0, J);   // rows J + 1 .. : both columns
            team.sync();
        }
        finish_column(J);
        if (!pair) break;
        update_cols(one{}, four{}, J + 1, J + 1, nb, J, J + 1);         // column J + 1: its last update, k = J
        team.sync();
        finish_column(J + 1);
    }
    return piv_min;
}

// The two triangular passes on a team of NW wavefronts (BigTeam): the same fma chains as big_solve below, each on one wavefront.
//   forward: per block column every wavefront solves the 16 x 16 unit triangle itself (same loads, same arithmetic — no broadcast through LDS, one barrier
//     per block column instead of two) and applies the block's 16 entries to ITS 64-row slots of the rows below (slot s of the block column -> wavefront
//     s mod NW); the panel entries of its slot in the NEXT block column are requested before the triangle solve;
//   backward: the column sums of a block (per lane one fma chain per column over its rows, slots ascending) are split by COLUMNS — wavefront w owns columns
//     4w .. 4w+3, i.e. two of the eight column pairs of every slab — so that every chain stays on one wavefront; the group sums and the block's own triangle run
//     on the first wavefront as in big_solve.
template <int NW>
__device__ __forceinline__ void big_solve_team(const double* LF, int N, int nb, int NPAD, double* v, double* bx, const BigTeam<NW>& team) {
    static_assert(NW == 4, "the backward pass deals the 16 columns of a block in quadruples");
    const int ln = lane_id();
    const int r16 = ln & 15;
    auto load_diag = [&](size_t o, double (&d)[16]) {
        const double* p = LF + o;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) { const big_d2 v2 = *reinterpret_cast<const big_d2*>(p + BigKkt::slab_pair(pp, r16)); d[2 * pp] = v2[0]; d[2 * pp + 1] = v2[1]; }
    };
    auto load_slot = [&](const double* pF, int J, int row0, double (&L)[16]) {
        const int row = row0 + ln;
        const int rw = (row < NPAD) ? row : NPAD - 1;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) { const big_d2 v2 = *reinterpret_cast<const big_d2*>(pF + BigKkt::slab_pair(pp, rw - 16 * J)); L[2 * pp] = v2[0]; L[2 * pp + 1] = v2[1]; }
    };
    // ---- forward
    {
        size_t oF = 0;
        double lrow[16], L[16];
        load_diag(0, lrow);
        load_slot(LF, 0, 16 + WAVE * team.w, L);
        double xprev = 0.0; int jprev = -1;   // the solved block of the previous step: stored behind that step's barrier (the other wavefronts were still reading the unsolved entries)
        for (int J = 0; J < nb; ++J) {
            const double* pF = LF + oF;
            const size_t oN = oF + BigKkt::sizeF(J, NPAD);
            double lnext[16], Lnext[16];
            if (J + 1 < nb) {   // requested before the dependent chains below: they depend on no solution entry
                load_diag(oN, lnext);
                const int rs = 16 * (J + 2) + WAVE * team.w;
                load_slot(LF + oN, J + 1, rs, Lnext);
            }
            if (team.lead() && jprev >= 0 && ln < 16 && 16 * jprev + r16 < N) v[16 * jprev + r16] = xprev;
            sched_fence();
            double xj[16];
            const int row = 16 * J + r16;
            double xr = (row < N) ? v[row] : 0.0;
#pragma unroll
            for (int c = 0; c < 15; ++c) {
                const double xc = bcast_lane(xr, c);
                const double up = fma(-lrow[c], xc, xr);
                xr = (r16 > c) ? up : xr;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) xj[c] = bcast_lane(xr, c);
            asm volatile("" :: "v"(lrow[15]));
            xprev = xr; jprev = J;
            bool first = true;
            for (int row0 = 16 * (J + 1) + WAVE * team.w; row0 < NPAD; row0 += WAVE * NW) {
                if (!first) load_slot(pF, J, row0, L);
                first = false;
                const int rw = row0 + ln;
                double vi = (rw < N) ? v[rw] : 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) vi = fma(-L[c], xj[c], vi);
                if (rw < N) v[rw] = vi;
            }
            team.sync_lds();
            if (J + 1 < nb) {
#pragma unroll
                for (int c = 0; c < 16; ++c) { lrow[c] = lnext[c]; L[c] = Lnext[c]; }
            }
            oF = oN;
        }
        if (team.lead() && jprev >= 0 && ln < 16 && 16 * jprev + r16 < N) v[16 * jprev + r16] = xprev;
        team.sync_lds();
    }
    // ---- diagonal
    for (int i = WAVE * team.w + ln; i < N; i += WAVE * NW) {
        const int I = i >> 4, r = i & 15;
        v[i] = v[i] / LF[BigKkt::offF(I, NPAD) + BigKkt::slab(r, r)];
    }
    team.sync_lds();
    // ---- backward
    double* red = bx + 16;
    double* grp = red + 16 * 65;
    size_t oFb = BigKkt::offF(nb, NPAD);
    constexpr int GB = 4;
    for (int J = nb - 1; J >= 0; --J) {
        oFb -= BigKkt::sizeF(J, NPAD);
        const double* pF = LF + oFb;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};   // columns 4w .. 4w + 3 of the block
        const int r = ln & 15, kq = ln >> 4;
        double lcol[16];
        if (team.lead()) {
#pragma unroll
            for (int c = 0; c < 16; ++c) lcol[c] = pF[BigKkt::slab(r, c)];
        }
        for (int row0 = 16 * (J + 1); row0 < NPAD; row0 += GB * WAVE) {
            big_d2 La[GB], Lb[GB]; double xr[GB];
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int row = row0 + g * WAVE + ln;
                const int rw = (row < NPAD) ? row : NPAD - 1;
                La[g] = *reinterpret_cast<const big_d2*>(pF + BigKkt::slab_pair(2 * team.w, rw - 16 * J));
                Lb[g] = *reinterpret_cast<const big_d2*>(pF + BigKkt::slab_pair(2 * team.w + 1, rw - 16 * J));
                xr[g] = (row < N) ? v[row] : 0.0;
            }
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                acc[0] = fma(La[g][0], xr[g], acc[0]); acc[1] = fma(La[g][1], xr[g], acc[1]);
                acc[2] = fma(Lb[g][0], xr[g], acc[2]); acc[3] = fma(Lb[g][1], xr[g], acc[3]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) red[(4 * team.w + c) * 65 + ln] = acc[c];
        team.sync_lds();
        if (team.lead()) {
            {
                double t[16], a = 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = red[r * 65 + 16 * kq + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) a += t[u];
                grp[kq * 16 + r] = a;
            }
            wsync();
            const int row = 16 * J + r;
            const double sum = (grp[r] + grp[16 + r]) + (grp[32 + r] + grp[48 + r]);
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
        team.sync_lds();
    }
}

// v <- K^{-1} v, v in LDS (N entries; padding rows are not touched). bx: unused LDS slots.
// SLIM: the build for TWO wavefronts per SIMD (256 registers, mid-size instances in batches above the SIMD count): two row slots per batch, no
// look-ahead — the look-ahead below holds 64 + 128 + 16 more registers and made that build spill (the 21-node robot grid: 29.3 -> 32.5 ms per 2048).
template <bool SLIM, int NW>
__device__ __forceinline__ void big_solve(const double* W, int N, double* v, double* bx, const BigTeam<NW>& team) {
    const int ln = lane_id();
    const int nb = BigKkt::nblk(N), NPAD = nb * 16;
    const size_t nt = (size_t)BigKkt::ntiles(N);
    const double* LF = W + nt * 256;
    if constexpr (NW > 1) { big_solve_team<NW>(LF, N, nb, NPAD, v, bx, team); return; }
#ifndef PMPC_BIG_GS
#define PMPC_BIG_GS 4
#endif
    constexpr int GS = SLIM ? 2 : PMPC_BIG_GS;   // row slots (of 64 rows) whose loads are issued together
    // ---- forward: blocks ascending; inside a block columns ascending. No entry of L depends on the solution, only the fma chains do: the panel of a block
    // column is requested BEFORE the triangular solve with its diagonal tile, and the diagonal tile of the next block column with it (two register sets,
    // alternating: the block loop is unrolled by two so that neither set is ever copied) — a block step exposed two memory round trips, now the first
    // one of the pass only.
    const int r16 = ln & 15;
    auto load_diag = [&](size_t o, double (&d)[16]) {
        const double* p = LF + o;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) { const big_d2 v2 = *reinterpret_cast<const big_d2*>(p + BigKkt::slab_pair(pp, r16)); d[2 * pp] = v2[0]; d[2 * pp + 1] = v2[1]; }
    };
    auto load_slots = [&](const double* pF, int J, int row0, double (&L)[GS][16]) {
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            const int row = row0 + g * WAVE + ln;
            const int rw = (row < NPAD) ? row : NPAD - 1;
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) { const big_d2 v2 = *reinterpret_cast<const big_d2*>(pF + BigKkt::slab_pair(pp, rw - 16 * J)); L[g][2 * pp] = v2[0]; L[g][2 * pp + 1] = v2[1]; }
        }
    };
    auto fwd_block = [&](int J, size_t oF, const double (&lrow)[16], double (&lnext)[16]) {
        const double* pF = LF + oF;
        double L[GS][16];
        if constexpr (!SLIM) {
            load_slots(pF, J, 16 * (J + 1), L);                                               // (a block column without rows below re-reads its last row: dropped)
            load_diag((J + 1 < nb) ? oF + BigKkt::sizeF(J, NPAD) : oF, lnext);                 // (the last block column re-reads its own tile: never used)
            sched_fence();
        }
        double xj[16];
        {   // finish x_J on 16 lanes (unit-lower triangular solve with the diagonal tile), then broadcast its 16 entries
            const int row = 16 * J + r16;
            double xr = (row < N) ? v[row] : 0.0;
#pragma unroll
            for (int c = 0; c < 15; ++c) {
                const double xc = bcast_lane(xr, c);
                const double up = fma(-lrow[c], xc, xr);
                xr = (r16 > c) ? up : xr;
            }
            if (ln < 16 && row < N) v[row] = xr;
#pragma unroll
            for (int c = 0; c < 16; ++c) xj[c] = bcast_lane(xr, c);
            asm volatile("" :: "v"(lrow[15]));   // (the unit diagonal is never read: without a use its register — the target of a 16-byte load — is handed out again while the load is in flight, and that write waits for EVERY load)
        }
        for (int row0 = 16 * (J + 1); row0 < NPAD; row0 += GS * WAVE) {
            if (SLIM || row0 != 16 * (J + 1)) load_slots(pF, J, row0, L);
#pragma unroll
            for (int g = 0; g < GS; ++g) {
                const int row = row0 + g * WAVE + ln;
                double vi = (row < N) ? v[row] : 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) vi = fma(-L[g][c], xj[c], vi);
                if (row < N) v[row] = vi;
            }
        }
        wsync();
    };
    constexpr int DG = SLIM ? 0 : 8;
    double dg[DG > 0 ? DG : 1];   // d_i of the rows 64 t + lane: four to eight loads whose round trips used to sit, one after the other, between the two passes
#pragma unroll
    for (int t = 0; t < DG; ++t) {
        const int i = 64 * t + ln;
        const int ic = (i < N) ? i : 0;
        dg[t] = LF[BigKkt::offF(ic >> 4, NPAD) + BigKkt::slab(ic & 15, ic & 15)];
    }
    {
        double dA[16], dB[16];
        load_diag(0, dA);
        size_t oF = 0;
        if constexpr (SLIM) {
            for (int J = 0; J < nb; ++J) { if (J > 0) load_diag(oF, dA); fwd_block(J, oF, dA, dB); oF += BigKkt::sizeF(J, NPAD); }
        } else {
            for (int J = 0; J < nb; J += 2) {
                fwd_block(J, oF, dA, dB);
                oF += BigKkt::sizeF(J, NPAD);
                if (J + 1 < nb) { fwd_block(J + 1, oF, dB, dA); oF += BigKkt::sizeF(J + 1, NPAD); }
            }
        }
    }
    // ---- diagonal (the first 512 entries were requested before the forward pass)
#pragma unroll
    for (int t = 0; t < DG; ++t) { const int i = 64 * t + ln; if (i < N) v[i] = v[i] / dg[t]; }
    for (int i = 64 * DG + ln; i < N; i += WAVE) {
        const int I = i >> 4, r = i & 15;
        v[i] = v[i] / LF[BigKkt::offF(I, NPAD) + BigKkt::slab(r, r)];
    }
    wsync();
    // ---- backward, from the SAME column panels (no row-ordered copy of L is read): per block of 16 columns, descending, the contributions of
    // the rows below the block are column dot products — every lane keeps one partial sum per column over its rows (row 16(J+1) + 64 g + lane,
    // g ascending, fma); the 64 partials of a column are added in four groups of 16 (lane (c, k) adds group k of column c in index order), the
    // group sums as (S0 + S1) + (S2 + S3) — subtracted once; then the block's own triangle, columns descending. (CPU restatement: PIVOT_BLOCKED.)
#ifndef PMPC_BIG_GB
#define PMPC_BIG_GB 4
#endif
    constexpr int GB = SLIM ? 2 : PMPC_BIG_GB;     // row slots in flight (16 running sums per lane on top of the loaded entries; 4: one batch per block column up to 272 rows)
    double* red = bx + 16;    // 16 x 64 partial sums, column stride 65 (conflict-free for the column-parallel group sums)
    double* grp = red + 16 * 65; // 4 x 16 group sums
    size_t oFb = BigKkt::offF(nb, NPAD);
    for (int J = nb - 1; J >= 0; --J) {
        oFb -= BigKkt::sizeF(J, NPAD);
        const double* pF = LF + oFb;
        double acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.0;
        const int r = ln & 15, kq = ln >> 4;
        double lcol[16];   // column r of the diagonal tile: L(16J + c, 16J + r), c = 0..15 — requested here, used behind the column sums (it depends on nothing)
        if constexpr (!SLIM) {
#pragma unroll
            for (int c = 0; c < 16; ++c) lcol[c] = pF[BigKkt::slab(r, c)];
        }
        for (int row0 = 16 * (J + 1); row0 < NPAD; row0 += GB * WAVE) {
            double L[GB][16], xr[GB];
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int row = row0 + g * WAVE + ln;
                const int rw = (row < NPAD) ? row : NPAD - 1;
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) { const big_d2 v2 = *reinterpret_cast<const big_d2*>(pF + BigKkt::slab_pair(pp, rw - 16 * J)); L[g][2 * pp] = v2[0]; L[g][2 * pp + 1] = v2[1]; }
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
            if constexpr (SLIM) {
#pragma unroll
                for (int c = 0; c < 16; ++c) lcol[c] = pF[BigKkt::slab(r, c)];
            }
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

// One solve of the condensed form on a team (boxadmm_solve, condensed mode):  t = r1 + A'(rho o r2);  x = S^{-1} t;  nu = rho o (A x - r2).
// rhs: [r1 | r2] in LDS, overwritten with [x | nu]; rho: the constraint penalties; scr: BigKkt::LDS_DOUBLES doubles of LDS (behind its first 256: the
// scaled r2, then the buffers of big_solve). The 64-entry chunks of the three element-wise / sparse-product loops are dealt round-robin.
template <bool SLIM, int NW, class JV>
__device__ __forceinline__ void big_cond_solve(const double* K, int n, int m, double* rhs, const double* rho, double* scr, const JV& jv, const BigTeam<NW>& team,
                                               long long* t_atu, long long* t_tri) {
    const int ln = lane_id();
    const long long t0 = t_atu ? clock64() : 0;
    double* u = scr + 256;
    for (int i0 = 0; i0 < m; i0 += WAVE) { if (!team.mine(i0 / WAVE)) continue; const int i = i0 + ln; if (i < m) u[i] = rho[i] * rhs[n + i]; }
    team.sync_lds();
    for (int c0 = 0; c0 < n; c0 += WAVE) {
        if (!team.mine(c0 / WAVE)) continue;
        const int c = c0 + ln;
        const typename JV::Col cc = jv.column(c < n ? c : 0);
        double bv[JV::NCB > 0 ? JV::NCB : 1];
        jv.col_block(cc, bv);
        double t;
        if constexpr ((int)JV::NG == 0 && (int)JV::NP == 0) t = jv.tab ? jv.coldot_fma_tab(cc, bv, u, rhs[c < n ? c : 0]) : jv.coldot_fma(cc, bv, u, rhs[c < n ? c : 0]);
        else t = jv.coldot_fma(cc, bv, u, rhs[c < n ? c : 0]);
        if (c < n) rhs[c] = t;
    }
    team.sync_lds();
    const long long s0 = t_atu ? clock64() : 0;
    big_solve<SLIM, NW>(K, n, rhs, scr + 256, team);
    const long long s1 = t_atu ? clock64() : 0;
    if (t_atu) { *t_atu += s0 - t0; *t_tri += s1 - s0; }
    for (int r0 = 0; r0 < m; r0 += WAVE) {
        if (!team.mine(r0 / WAVE)) continue;
        const int r = r0 + ln;
        const typename JV::Row rw = jv.rowinfo(r < m ? r : 0);
        double bv[JV::NDER];
        jv.row_block(rw, bv);
        double a;
        if constexpr ((int)JV::NG == 0 && (int)JV::NP == 0) a = jv.tab ? jv.rowdot_fma_tab(rw, bv, rhs) : jv.rowdot_fma(rw, bv, rhs);
        else a = jv.rowdot_fma(rw, bv, rhs);
        if (r < m) rhs[n + r] = rho[r] * (a - rhs[n + r]);
    }
    team.sync_lds();
}

// Mailbox of a four-wavefront team in LDS (BigTeam): the first wavefront posts a routine, the workgroup barrier behind the post releases the helpers, every
// routine ends on a barrier of its own; the helpers then wait for the next post.
constexpr int BIG_MAIL_DOUBLES = 60;
template <class Model> struct JViewRT;   // pmpc_jview.hpp
template <class JV>
struct BigMail {
    int op, n, m, ldh;
    double* K; const double* H; const double* kdiag; const double* rho; double* rhs; double* scr;
    double red4[4];
    const double* var; const double* lam;   // BIG_OP_STAGE1 / 2: the iterate and the multipliers of the AD stages
    double* Hw; const double* pa[5]; double* pw[2]; double f[2]; int ldw, fast;   // BIG_OP_BFGS_*: the workspace, (step, lgn, lg | Bs, r), (Bs, y), (sBs, sr)
    JV jv;
};
template <class Model, class OcpT>
__device__ __forceinline__ void big_helper_loop(BigMail<JViewRT<Model>>* mb, int wv, OcpT& ocp) {
    using JV = JViewRT<Model>;
    static_assert(sizeof(BigMail<JV>) <= BIG_MAIL_DOUBLES * sizeof(double), "mailbox fits its LDS slot");
    BigTeam<4> team; team.w = wv;
    for (;;) {
        __syncthreads();
        const int op = __builtin_amdgcn_readfirstlane(mb->op);
        if (op == BIG_OP_EXIT) return;
        const int n = __builtin_amdgcn_readfirstlane(mb->n), m = __builtin_amdgcn_readfirstlane(mb->m);
        if (op == BIG_OP_STAGE2) {   // the entry-per-lane second-order AD stage of the exact linearisation: 4096 independent entries on config C
            ocp.template stage_second_order_entry_part<4>(mb->var, mb->lam, wv);
            team.sync();
        } else if (op == BIG_OP_STAGE1) {   // the (node, direction) pairs of the first-order AD stage
            ocp.template stage_first_order_part<4>(mb->var, wv);
            team.sync();
        } else if (op == BIG_OP_BFGS_BS) {   // B s and y, row per lane
            bfgs_rows_products<4, BIG_MEM_BATCH>(mb->Hw, __builtin_amdgcn_readfirstlane(mb->ldw), n, mb->pa[0], mb->pa[1], mb->pa[2], mb->pw[0], mb->pw[1], wv);
            team.sync();
        } else if (op == BIG_OP_BFGS_R2) {   // the rank-2 update, row per lane
            if (__builtin_amdgcn_readfirstlane(mb->fast)) bfgs_rows_rank2<4, BIG_MEM_BATCH, true>(mb->Hw, __builtin_amdgcn_readfirstlane(mb->ldw), n, mb->pa[3], mb->pa[4], mb->f[0], mb->f[1], wv);
            else bfgs_rows_rank2<4, BIG_MEM_BATCH, false>(mb->Hw, __builtin_amdgcn_readfirstlane(mb->ldw), n, mb->pa[3], mb->pa[4], mb->f[0], mb->f[1], wv);
            team.sync();
        } else if (op == BIG_OP_FACTOR) {
            (void)big_build_condensed<JV, 4>(mb->K, n, m, mb->H, __builtin_amdgcn_readfirstlane(mb->ldh), mb->kdiag, mb->rho, mb->jv, team, mb->red4);
            (void)big_factor<4>(mb->K, n, mb->scr, team);
        } else {
            big_cond_solve<false, 4, JV>(mb->K, n, m, mb->rhs, mb->rho, mb->scr, mb->jv, team, nullptr, nullptr);
        }
    }
}

}  // namespace pmpc
