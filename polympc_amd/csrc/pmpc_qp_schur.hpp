// polympc_amd — box-ADMM QP solve for a Hessian that is BLOCK DIAGONAL PER COLLOCATION NODE (one wavefront per QP).
//
// Every control test of the reference plugs the problem's sparsity-preserving block BFGS into the solver
// (continuous_ocp.hpp:2304-2431; cstr_control_test.cpp:128-132, mpc_wrapper_test.cpp:100-105, minimal_time_test.cpp:84-88,
// valet_parking_mpc_test.cpp:161-165), and the exact Lagrangian Hessian is block diagonal by construction (:2128-2173): H then consists of one
// (NX+NU) x (NX+NU) block per node. The reference's boxADMM still factorises the (n+m)-row KKT matrix of box_admm.hpp:209-223 — with SimplicialLDLT in
// the SPARSE tests, which is where this structure pays. Here:
//     K = [ P  A' ; A  -1/rho ],   P = H + sigma I + rho_box  (block diagonal),   A = J (collocation Jacobian)
//   * factorisation:  Q_k = P_k^{-1} per node (one lane per node, symmetric sweeps on d = NX+NU pivots);  S = 1/rho + A Q A'  (m x m, m = NX nn <= 64,
//     one row per lane, formed from the per-node blocks of J and the differentiation matrix in LDS — A is never dense);  W = -S^{-1} by the blocked
//     sweep on the matrix cores of pmpc_qp_reg.hpp (RegKkt<m>: 21 / 44 / 48 pivots instead of 56 / 110 / 128)
//   * solve:  t = Q r1,  g = A t - r2,  nu = S^{-1} g (the DPP mat-vec of RegKkt),  x = Q (r1 - A' nu),  then ONE step of iterative refinement on the
//     constraint rows (e = A x - nu/rho - r2,  nu += S^{-1} e,  x = Q (r1 - A' nu)): the swept inverse of S has an isotropic forward error
//     ~ eps cond(S) |nu| while x tolerates errors of nu only in the near-null directions of Q^(1/2) A'; with the step the solve is MORE accurate than the
//     dense orders (config B after a rho update, cond(S) = 6e5: |dx| 6e-13 against 8e-12; without it 2e-8) — DESIGN.md §4.
//   * every ADMM vector is one register per lane and slot: primal entries g = lane + 64 e (e < SLOTS, n <= 128), constraint rows on lanes [0, m).
//     Block products go through LDS (the d entries of a node), the sparse products with A are fma chains over the own-node block and ALL nodes of
//     the grid with the coefficient 0 outside the row's / column's segment — no divergence, and the same statement for non-finite operands.
//   * ONE PARAMETER (NP = 1, round 5: minimal_time_test.cpp's problem): H has the arrow shape — node blocks, a border row / column, a corner — and A a dense
//     column a_p. The parameter is the last primal entry (one more lane); with K0 the matrix above on the node variables, w = [H(p, z); a_p] and
//     pi = H_pp + sigma + rho_p:  factorisation (q_z, q_nu) = K0^{-1} w, delta = pi - w'q;  solve (z0, nu0) = K0^{-1} [r1_z; r2],
//     p = (r1_p - w'[z0; nu0]) / delta,  z = z0 - p q_z,  nu = nu0 - p q_nu  (fma). w'v: one fma chain per lane (primal slots ascending, then the
//     constraint row), the 64 partial sums added pairwise over adjacent lanes by the DPP tree of wave_sum (pmpc_qp.hpp).
//   * residuals (box_admm.hpp:398-415): H x from the blocks, A x / A' y from pmpc_jview.hpp — the non-zero products of the reference's dense chains in
//     the same ascending order (multiply, then add).
// The CPU restatement of exactly this order is PIVOT_SCHUR (the CPU checker of the test suite): the kernels are checked bit for bit against it, and it is tied to the
// reference's pivoted LDL^T on the QP streams of every configuration by the CPU tests.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_jview.hpp"
#include "pmpc_qp_reg.hpp"

namespace pmpc {

template <class Model, int PP, int SS>
struct SchurDims {
    enum { NX = Model::NX, NU = Model::NU, NPAR = Model::NP, D = NX + NU, DD = D * D, JBS = OcpDims<Model>::JBS /* row stride of jblk (odd) */, NNODES = PP * SS + 1,
           N0 = D * NNODES /* node variables */, N = N0 + NPAR, M = NX * NNODES, VARX = NX * NNODES,
           SLOTS = (N + WAVE - 1) / WAVE, PE = N0 / WAVE, PL = N0 % WAVE /* slot and lane of the parameter (NPAR = 1) */, P1 = PP + 1, NNR = NNODES + (NNODES & 1),   // entries read per table row (two at a time)
           NNP = lds_row_stride(NNODES),                                                // row stride: even, NNP / 2 odd — bank-conflict-free per-lane row bases (pmpc_jview.hpp)
           TAB = NNODES * NNP + (NNODES + 1) * NNP };
    static_assert(Model::NP <= 1 && Model::NG == 0, "block-structured QP: at most one parameter (bordered form), no path constraints");
    static_assert(M <= WAVE && N <= 2 * WAVE, "block-structured QP: at most 64 constraint rows and 128 variables");
    // segment start / D row of the equality rows of node k (Ocp::seg_row) and the structural coupling of row node r with column node k
    __host__ __device__ static constexpr bool last(int k) { return k == NNODES - 1; }
    __host__ __device__ static constexpr int kb(int k) { return last(k) ? NNODES - 1 - PP : (k / PP) * PP; }
    __host__ __device__ static constexpr int row(int k) { return last(k) ? PP : k % PP; }
    __host__ __device__ static constexpr bool coupled(int r, int k) { return k != r && k >= kb(r) && k <= kb(r) + PP; }
    // index of D~(r, k) in OcpLds::D (the last node's row -D(0, P - t) is stored behind the matrix)
    __host__ __device__ static constexpr int dti(int r, int k) { return last(r) ? P1 * P1 + (k - kb(r)) : row(r) + (k - kb(r)) * P1; }
    // D~ as two dense tables in LDS, built once per kernel (schur_build_tables): Dt[r NNP + k] = D~(r, k) — the differentiation-matrix entry of
    // equality row node r on column node k, 0 on the own node (its entry lives in the node block) and outside the row's segment — and its transpose
    // DtT[k NNP + r] followed by one all-zero row (read by the control columns). Rows are contiguous: 16-byte reads.
    // LDS doubles the solver needs beside the staging of RegKkt<M>: Q blocks, one primal and one dual exchange vector, the tables
    static constexpr int LDS_DOUBLES = NNODES * DD + (N + 1) + (M + 1) + TAB + 1;   // (round 5: the KKT diagonal lives in a register per primal slot and visits the primal exchange vector for the factorisation — 81 doubles less on the 16-node grid, whose instance then fits a quarter of a CU's LDS)
};

// boxADMM::solve_impl (7-argument form: zero guesses, box_admm.hpp:81-86) on the block structure. hblk: [k][c' * D + c] = H(g(k, c), g(k, c')), the
// node blocks of H column-major (LOWER triangle read for the KKT matrix, as Eigen::LDLT does; the full block for H x); jblk: [(k NX + q) D + c] =
// J(k NX + q, g(k, c)); Dm: OcpLds::D; nsr: the node table of Ocp::stage_constants (pmpc_jview.hpp reads it); h / bounds: LDS vectors.
// tr: RegKkt<M>::TRI doubles of staging; qblk / xsc / dsc: NNODES D^2 / N + 1 / M + 1 doubles of LDS that live through the solve.
// builds the two D~ tables (see SchurDims) from the differentiation matrix in LDS; once per kernel — the structure does not depend on the iterate
template <class Model, int PP, int SS>
__device__ __forceinline__ void schur_build_tables(const double* Dm, double* Dt) {
    using SD = SchurDims<Model, PP, SS>;
    constexpr int NNODES = SD::NNODES, NNP = SD::NNP;
    double* DtT = Dt + NNODES * NNP;
    for (int e = lane_id(); e < SD::TAB; e += WAVE) Dt[e] = 0.0;
    wsync();
    for (int e = lane_id(); e < NNODES * NNODES; e += WAVE) {
        const int r = e / NNODES, k = e - r * NNODES;
        const bool lastr = r == NNODES - 1;
        const int kbr = lastr ? NNODES - 1 - PP : (r / PP) * PP;
        const int rowr = lastr ? PP : r - kbr;
        const int t = k - kbr;
        const bool cpl = k != r && (unsigned)t <= (unsigned)PP;
        const double dv = Dm[lastr ? SD::P1 * SD::P1 + (cpl ? t : 0) : rowr + (cpl ? t : 0) * SD::P1];
        const double v = cpl ? dv : 0.0;
        Dt[r * NNP + k] = v;
        DtT[k * NNP + r] = v;
    }
    wsync();
}

// boxADMM::solve_impl (7-argument form: zero guesses, box_admm.hpp:81-86) on the block structure. hblk: [k][c' * D + c] = H(g(k, c), g(k, c')), the
// node blocks of H column-major (LOWER triangle read for the KKT matrix, as Eigen::LDLT does; the full block for H x); jblk: [(k NX + q) D + c] =
// J(k NX + q, g(k, c)); Dm: OcpLds::D; nsr: the node table of Ocp::stage_constants (pmpc_jview.hpp reads it); h / bounds: LDS vectors.
// tr: RegKkt<M>::TRI doubles of staging; qblk / xsc / dsc: NNODES D^2 / N + 1 / M + 1 doubles of LDS that live through the solve;
// Dt: the tables of schur_build_tables. hbrd (NP = 1): [H(p, 0..N0-1), H(p, p) | H(0..N0-1, p)] — the border row with the corner, then the border column
// (the KKT matrix reads the row: lower triangle; H x reads both).
template <class Model, int PP, int SS>
__device__ __forceinline__ void boxadmm_solve_schur(const double* hblk, const double* hbrd, const double* h, const double* jblk, const double* Dm, const int* nsr,
                                                    const double* Alb, const double* Aub, const double* xlb, const double* xub,
                                                    const pmpc_qp_settings& s, pmpc_qp_info& info, double* out_x, double* out_y, double* tr,
                                                    double* qblk, double* xsc, double* dsc, const double* Dt, long long* dbg = nullptr,
                                                    long long* tm = nullptr) {
    using SD = SchurDims<Model, PP, SS>;
    constexpr int NX = SD::NX, NU = SD::NU, D = SD::D, DD = SD::DD, NNODES = SD::NNODES, N = SD::N, N0 = SD::N0, NPAR = SD::NPAR, M = SD::M, VARX = SD::VARX, SLOTS = SD::SLOTS, NNP = SD::NNP, NNR = SD::NNR;
    using d2 = double __attribute__((ext_vector_type(2)));
    const long long tp0 = dbg ? clock64() : 0;
    // ---- lane roles -------------------------------------------------------------------------------------------------------------------------
    // primal slot e: entry g = lane + 64 e of [x_0 .. x_{nn-1} | u_0 .. u_{nn-1}] = (node k, block position c); clamped duplicates beyond n.
    // Stores of a lane without an entry go to a dummy slot behind the vector (xsc[N], dsc[M]) instead of through a divergent branch: VGPR
    // spills inside partial-EXEC regions lose the inactive lanes' copies (DESIGN.md compiler hazard 3). The index arithmetic is a function of the
    // lane id alone; the blocks that need more of it than the ADMM loop (factorisation, residuals) re-derive it there (`roles`) instead of keeping a
    // dozen integers alive through the loop.
    // (NP = 1: the parameter is entry N0; it — and the clamped duplicates behind it — take node 0's addresses with all block coefficients zero: isp)
    struct Role { bool pv, isp; int g, px, k, c, xb, ub; };
    auto role = [](int e) -> Role {
        Role r;
        const int g = lane_id() + WAVE * e;
        r.pv = g < N; r.g = r.pv ? g : N - 1; r.px = r.pv ? g : N;
        r.isp = NPAR > 0 && r.g >= N0;
        const int gz = r.isp ? 0 : r.g;
        const bool isx = gz < VARX;
        r.k = isx ? gz / NX : (gz - VARX) / NU;
        r.c = isx ? gz - r.k * NX : NX + (gz - VARX) - r.k * NU;
        r.xb = r.k * NX; r.ub = VARX + r.k * NU;
        return r;
    };
    const int ln = lane_id();
    const bool isC = ln < M;                // constraint row ci = lane (clamped duplicates beyond m) = (node ni, state si)
    const int ci = isC ? ln : M - 1;
    const int pc = isC ? ln : M;
    const int ni = ci / NX, si = ci - ni * NX;

    // ---- per-lane problem data that the ADMM loop keeps in registers ---------------------------------------------------------------------------
    double hv[SLOTS], lo[SLOTS], hi[SLOTS], rhob[SLOTS], rhobinv[SLOTS]; int typ[SLOTS];
    double kdv[SLOTS];                      // diagonal of P = H + sigma I + rho_box per primal slot (construct_kkt_matrix / update_kkt_rho: carried, updated incrementally)
    double colb[SLOTS][NX];                 // column g of A inside its own node's rows
    double qz[SLOTS], qnu = 0.0, delta = 1.0; bool ispE[SLOTS];   // NP = 1: K0^{-1} w, the pivot of the border
    // the border itself — H(p, g) per primal slot (0 on the parameter's lanes) and the parameter's column of A on this row (0 beyond the rows) — is re-read from LDS
    // where it is used (the opaque zero keeps the reads there): two more doubles alive across the swept inverse put a scratch reload inside it (ISA test)
    auto border_row = [&](int e, int zo) -> double { const Role r = role(e); const double br = hbrd[zo + (r.isp ? 0 : r.g)]; return r.isp ? 0.0 : br; };
    auto border_col = [&](int zo) -> double { const double a = jblk[zo + ci * SD::JBS + D]; return isC ? a : 0.0; };
    const double* xo[SLOTS]; const double* uo[SLOTS]; const double* dcol[SLOTS]; const double* nuo[SLOTS]; const double* nuc[SLOTS]; double* xst[SLOTS];
    double rho = s.rho;
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) {
        const Role r = role(e);
        hv[e] = h[r.g]; lo[e] = xlb[r.g]; hi[e] = xub[r.g];
        typ[e] = classify_bounds(lo[e], hi[e]);
        rhob[e] = rho_of(typ[e], rho); rhobinv[e] = 1.0 / rhob[e];
        double kd = hblk[r.k * DD + r.c * D + r.c];
        ispE[e] = r.isp; qz[e] = 0.0;
        if constexpr (NPAR > 0) { const double kc = hbrd[N0]; kd = r.isp ? kc : kd; }
        kd += s.sigma; kd += rhob[e];   // construct_kkt_matrix, box_admm.hpp:214-216
        kdv[e] = kd;
#pragma unroll
        for (int q = 0; q < NX; ++q) { const double cb = jblk[(r.k * NX + q) * SD::JBS + r.c]; colb[e][q] = r.isp ? 0.0 : cb; }
        xo[e] = xsc + r.xb; uo[e] = xsc + r.ub; xst[e] = xsc + r.px;
        dcol[e] = Dt + NNODES * NNP + ((r.c < NX && !r.isp) ? r.k : NNODES) * NNP;   // column node k of D~ (the all-zero row for a control column and the parameter)
        nuo[e] = dsc + r.xb; nuc[e] = dsc + (r.c < NX ? r.c : 0);
    }
    const double loA = Alb[ci], hiA = Aub[ci];
    const int typA = classify_bounds(loA, hiA);
    double rhoA = rho_of(typA, rho), rinvA = 1.0 / rhoA;
    double bi[D];                           // row ci of A inside its own node's columns
#pragma unroll
    for (int c = 0; c < D; ++c) bi[c] = jblk[ci * SD::JBS + c];
    const double* drow = Dt + ni * NNP;     // row node ni of D~
    const double* xoi = xsc + ni * NX; const double* uoi = xsc + VARX + ni * NU; const double* xsi = xsc + si;
    double* dst = dsc + pc;

    // ---- state: x, q, y_box per primal slot; z, y_a on the constraint lanes (zero guesses) ---------------------------------------------------
    double xv[SLOTS], qv[SLOTS], yb[SLOTS];
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) { xv[e] = 0.0; qv[e] = 0.0; yb[e] = 0.0; }
    double zv = 0.0, ya = 0.0;

    RegKkt<M> K;
    double Qrow[SLOTS][D];
    constexpr int GAVE_UP = 100;   // internal status: the conditioning gate tripped at a factorisation (reported as UNSOLVED + PMPC_FLAG_ILLCOND)
    int status = PMPC_QP_UNSOLVED, rho_updates = 1;
    const double alpha = s.alpha;
    double max_Ax_z_norm = 0.0, max_Hx_ATy_h_norm = 0.0, res_prim = 1.0, res_dual = 1.0, rho_estimate = 0.0;
    int iter = 1;
    int until_check = s.check_termination, until_adapt = s.adaptive_rho_interval;
    bool running = true;
    if (dbg) dbg[17] += clock64() - tp0;

    // sum_c' fma(Qrow[c'], v[g(k, c')], .) for the node of slot e, v = xsc (c' ascending from 0)
    auto qprod = [&](int e) -> double {
        double vx[D];
#pragma unroll
        for (int c = 0; c < NX; ++c) vx[c] = xo[e][c];
#pragma unroll
        for (int c = 0; c < NU; ++c) vx[NX + c] = uo[e][c];
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) a = fma(Qrow[e][c], vx[c], a);
        return a;
    };
    // (A v)_ci, v = xsc: own block fma chain over c ascending, then every node kk ascending with the row's D~ coefficient (0 outside its segment and
    // on the own node), read from the table two at a time
    auto arow = [&]() -> double {
        double vo[D], vn[NNODES]; d2 dc[NNR / 2];
#pragma unroll
        for (int c = 0; c < NX; ++c) vo[c] = xoi[c];
#pragma unroll
        for (int c = 0; c < NU; ++c) vo[NX + c] = uoi[c];
#pragma unroll
        for (int kk = 0; kk < NNR / 2; ++kk) dc[kk] = *reinterpret_cast<const d2*>(drow + 2 * kk);
#pragma unroll
        for (int kk = 0; kk < NNODES; ++kk) vn[kk] = xsi[kk * NX];
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) a = fma(bi[c], vo[c], a);
#pragma unroll
        for (int kk = 0; kk < NNODES; ++kk) a = fma(dc[kk / 2][kk & 1], vn[kk], a);
        return a;
    };
    // (A' nu)_g for slot e, nu = dsc: the own node's rows q ascending, then every row node kr ascending
    auto acol = [&](int e) -> double {
        double vo[NX], vn[NNODES]; d2 dc[NNR / 2];
#pragma unroll
        for (int q = 0; q < NX; ++q) vo[q] = nuo[e][q];
#pragma unroll
        for (int kr = 0; kr < NNR / 2; ++kr) dc[kr] = *reinterpret_cast<const d2*>(dcol[e] + 2 * kr);
#pragma unroll
        for (int kr = 0; kr < NNODES; ++kr) vn[kr] = nuc[e][kr * NX];
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < NX; ++q) a = fma(colb[e][q], vo[q], a);
#pragma unroll
        for (int kr = 0; kr < NNODES; ++kr) a = fma(dc[kr / 2][kr & 1], vn[kr], a);
        return a;
    };

    // K0^{-1} [r1; r2] on the node variables: the range-space solve and one step of iterative refinement on the constraint rows (sol, nu)
    auto solve0 = [&](const double (&r1)[SLOTS], const double r2, double (&sol)[SLOTS], double& nu) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < SLOTS; ++e) *xst[e] = r1[e];
        lds_order();
        double tt[SLOTS];
#pragma unroll
        for (int e = 0; e < SLOTS; ++e) tt[e] = qprod(e);
        lds_order();
#pragma unroll
        for (int e = 0; e < SLOTS; ++e) *xst[e] = tt[e];
        lds_order();
        const double g1 = arow() - r2;
        nu = K.apply(isC ? g1 : 0.0);
        lds_order();
        *dst = nu;
        lds_order();
#pragma unroll
        for (int e = 0; e < SLOTS; ++e) { const double u = r1[e] - acol(e); *xst[e] = u; }
        lds_order();
#pragma unroll
        for (int e = 0; e < SLOTS; ++e) sol[e] = qprod(e);
        lds_order();
        // one step of iterative refinement on the constraint rows
#pragma unroll
        for (int e = 0; e < SLOTS; ++e) *xst[e] = sol[e];
        lds_order();
        const double e1 = fma(-rinvA, nu, arow()) - r2;
        const double dnu = K.apply(isC ? e1 : 0.0);
        nu = nu + dnu;
        lds_order();
        *dst = nu;
        lds_order();
#pragma unroll
        for (int e = 0; e < SLOTS; ++e) { const double u = r1[e] - acol(e); *xst[e] = u; }
        lds_order();
#pragma unroll
        for (int e = 0; e < SLOTS; ++e) sol[e] = qprod(e);
        lds_order();
    };

    while (running) {
        {   // ---- factorisation: Q_k, S, W = -S^{-1} (construct_kkt_matrix + factorise_kkt_matrix, box_admm.hpp:209-223, :336-341) ---------------
            const long long f0 = dbg ? clock64() : 0;
            // opaque zero on the LDS bases of this block: its reads (node blocks, differentiation matrix) are invariant across the factorisation loop and
            // would otherwise be hoisted in front of it and kept — or spilled — through every ADMM iteration (DESIGN.md compiler hazard 1)
            int zf = 0; asm volatile("" : "+v"(zf));
            const double* hbF = hblk + zf; const double* jbF = jblk + zf; const double* DmF = Dm + zf; double* qbF = qblk + zf; const double* pdF = xsc + zf;
            // the KKT diagonal visits the primal exchange vector (free between ADMM iterations): the per-node inverses read it across lanes
#pragma unroll
            for (int e = 0; e < SLOTS; ++e) *xst[e] = kdv[e];
            lds_order();
            const double pi_border = NPAR > 0 ? pdF[NPAR > 0 ? N0 : 0] : 0.0;   // (read before the border's inner solve reuses the vector)
            const double* DtF = Dt + zf;
            {   // one lane per node (clamped duplicates beyond the last node compute and store the last node's block again)
                const int k = ln < NNODES ? ln : NNODES - 1;
                double Mx[D][D];   // lower triangle used: Mx[i][j], i >= j
#pragma unroll
                for (int j = 0; j < D; ++j)
#pragma unroll
                    for (int i = j; i < D; ++i) {
                        const int gi = i < NX ? k * NX + i : VARX + k * NU + (i - NX);
                        Mx[i][j] = (i == j) ? pdF[gi] : hbF[k * DD + j * D + i];
                    }
#pragma unroll
                for (int p = 0; p < D; ++p) {
                    const double r = 1.0 / Mx[p][p];
                    double c[D], l[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) { c[i] = (i >= p) ? Mx[i][p] : Mx[p][i]; l[i] = c[i] * r; }
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        if (j == p) continue;
#pragma unroll
                        for (int i = j; i < D; ++i) { if (i == p) continue; Mx[i][j] = fma(-l[i], c[j], Mx[i][j]); }
                    }
#pragma unroll
                    for (int i = 0; i < D; ++i) { if (i == p) continue; if (i >= p) Mx[i][p] = l[i]; else Mx[p][i] = l[i]; }
                    Mx[p][p] = -r;
                }
                lds_order();
#pragma unroll
                for (int j = 0; j < D; ++j)
#pragma unroll
                    for (int i = 0; i < D; ++i) qbF[k * DD + j * D + i] = -((i >= j) ? Mx[i][j] : Mx[j][i]);
            }
            lds_order();
            if (dbg) dbg[15] += clock64() - f0;   // (phase-profile build: node-block inverses; shares the slot of the line-search prologue)
            const long long f1 = dbg ? clock64() : 0;
#pragma unroll
            for (int e = 0; e < SLOTS; ++e) {
                const Role r = role(e);
#pragma unroll
                for (int c = 0; c < D; ++c) { const double qv_ = qbF[r.k * DD + c * D + r.c]; Qrow[e][c] = r.isp ? 0.0 : qv_; }
            }
            // rows ci of G = A Q and of S = 1/rho + G A', node by node (kk ascending): g = row ci of G on the columns of node kk — own node: fma chains
            // over the block; any other node: (D~ coefficient, 0 when uncoupled) * Q_kk(si, .) — then every S(ci, j) that node kk enters: the rows j of
            // node kk itself take the chain over their own block, the rows of the nodes coupled to kk one product with their D~ entry
            double srow[M];
#pragma unroll
            for (int j = 0; j < M; ++j) srow[j] = (ci == j) ? rinvA : 0.0;
            // (every group of LDS reads below is issued as ONE batch in front of the arithmetic that consumes it — a scheduling fence between the two:
            //  left to itself the compiler pairs each read with its fma and waits a full LDS round trip per entry, 70 k cycles per factorisation)
            double own[D];
            {
                double qo[D][D];
#pragma unroll
                for (int cc = 0; cc < D; ++cc)
#pragma unroll
                    for (int c2 = 0; c2 < D; ++c2) qo[cc][c2] = qbF[ni * DD + cc * D + c2];
                sched_fence();
#pragma unroll
                for (int cc = 0; cc < D; ++cc) {
                    double a = 0.0;
#pragma unroll
                    for (int c2 = 0; c2 < D; ++c2) a = fma(bi[c2], qo[cc][c2], a);
                    own[cc] = a;
                }
            }
#pragma unroll
            for (int kk = 0; kk < NNODES; ++kk) {
                double qk[D], jb[NX][D], dco[NNODES];
                const double dr = DtF[ni * NNP + kk];
#pragma unroll
                for (int cc = 0; cc < D; ++cc) qk[cc] = qbF[kk * DD + cc * D + si];
#pragma unroll
                for (int q = 0; q < NX; ++q)
#pragma unroll
                    for (int cc = 0; cc < D; ++cc) jb[q][cc] = jbF[(kk * NX + q) * SD::JBS + cc];      // own blocks of the rows of node kk (wave-uniform addresses)
#pragma unroll
                for (int nj = 0; nj < NNODES; ++nj) dco[nj] = SD::coupled(nj, kk) ? DmF[SD::dti(nj, kk)] : 0.0;   // D~(nj, kk) of the row nodes coupled to kk
                sched_fence();
                double g[D];
#pragma unroll
                for (int cc = 0; cc < D; ++cc) { const double pr = dr * qk[cc]; g[cc] = (kk == ni) ? own[cc] : pr; }
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int nj = j / NX, sj = j - nj * NX;
                    if (nj == kk) {
#pragma unroll
                        for (int cc = 0; cc < D; ++cc) srow[j] = fma(g[cc], jb[sj][cc], srow[j]);
                    } else if (SD::coupled(nj, kk)) srow[j] = fma(g[sj], dco[nj], srow[j]);
                }
                sched_fence();
            }
            double diag = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) diag = (ci == j) ? srow[j] : diag;
            sched_fence();
            if (dbg) dbg[16] += clock64() - f1;   // (phase-profile build: rows of G and S; shares the slot of the line-search acceptance)
            // Conditioning gate (PMPC_FLAG_ILLCOND, round 5): cond(S) = rho_eq lambda_max(A Q A') once 1/rho is what keeps S regular — the state columns of a
            // collocation Jacobian are nearly singular when the dynamics hardly depend on the states (D has the constant profile in its null space), and the
            // bounded controls' Q ~ 1/rho closes the gap less and less as rho adapts upwards: the error of the range-space solve is ~ eps cond(S), whatever
            // the refinement step does (its residual is evaluated in working precision). Beyond PMPC_SCHUR_COND_GATE the QP is given up; the SQP kernel ends
            // the instance (PMPC_SQP_REDO) and the launcher's redo launch solves it in the (n + m)-row form. None on any BASELINE workload.
            const bool tripped = K.template invert<M, true>(ln, tr, diag, [&](int j, int) -> double { return srow[j < M ? j : 0]; }, tm, 0.0, typename RegKkt<M>::NoPre(), PMPC_SCHUR_COND_GATE);
            if (tripped) { status = GAVE_UP; running = false; if (dbg) dbg[0] += clock64() - f0; break; }
            if constexpr (NPAR > 0) {   // border: (q_z, q_nu) = K0^{-1} [H(p, z); a_p],  delta = (H_pp + sigma + rho_p) - w'q
                double sq[SLOTS], nq;
                double bz[SLOTS];
#pragma unroll
                for (int e = 0; e < SLOTS; ++e) bz[e] = border_row(e, zf);
                const double ap = border_col(zf);
                solve0(bz, ap, sq, nq);
                double part = 0.0;
#pragma unroll
                for (int e = 0; e < SLOTS; ++e) { sq[e] = ispE[e] ? 0.0 : sq[e]; part = fma(bz[e], sq[e], part); }
                nq = isC ? nq : 0.0;
                part = fma(ap, nq, part);
                delta = pi_border - wave_sum(part);
#pragma unroll
                for (int e = 0; e < SLOTS; ++e) qz[e] = ispE[e] ? -1.0 : sq[e];   // (z - p q_z leaves p itself on the parameter's lane)
                qnu = nq;
            }
            if (dbg) dbg[0] += clock64() - f0;
        }
        bool refactor = false;
        for (; iter <= s.max_iter; ++iter) {
            const long long ti0 = dbg ? clock64() : 0;
            // compute_kkt_rhs, box_admm.hpp:351-355
            double r1[SLOTS];
#pragma unroll
            for (int e = 0; e < SLOTS; ++e) r1[e] = ((s.sigma * xv[e] - hv[e]) + rhob[e] * qv[e]) - yb[e];
            const double r2 = zv - rinvA * ya;
            // range-space solve (+ the border of the parameter)
            double sol[SLOTS], nu;
            solve0(r1, r2, sol, nu);
            if constexpr (NPAR > 0) {
                int zb = 0; asm volatile("" : "+v"(zb));
                double part = 0.0;
#pragma unroll
                for (int e = 0; e < SLOTS; ++e) { sol[e] = ispE[e] ? 0.0 : sol[e]; part = fma(border_row(e, zb), sol[e], part); }
                part = fma(border_col(zb), isC ? nu : 0.0, part);
                const double dot = wave_sum(part);
                const double r1p = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r1[SD::PE]), SD::PL), __builtin_amdgcn_readlane(__double2loint(r1[SD::PE]), SD::PL));
                const double pval = (r1p - dot) / delta;
#pragma unroll
                for (int e = 0; e < SLOTS; ++e) sol[e] = fma(-pval, qz[e], sol[e]);
                nu = fma(-pval, qnu, nu);
            }
            // ADMM updates, box_admm.hpp:125-147 (Q1: x = alpha x~; x += (1 - alpha) x)
            {
                const double zprev = zv;
                const double zt = zprev + rinvA * (nu - ya);
                double zz = alpha * zt;
                zz += (1 - alpha) * zprev + rinvA * ya;
                zz = fmin(fmax(zz, loA), hiA);
                ya = ya + rhoA * ((alpha * zt + (1 - alpha) * zprev) - zz);
                zv = zz;
            }
#pragma unroll
            for (int e = 0; e < SLOTS; ++e) {
                double xx = alpha * sol[e];
                xx += (1 - alpha) * xx;
                double qq = xx + rhobinv[e] * yb[e];
                qq = fmin(fmax(qq, lo[e]), hi[e]);
                yb[e] = yb[e] + rhob[e] * (xx - qq);
                xv[e] = xx; qv[e] = qq;
            }
            if (dbg) dbg[9] += clock64() - ti0;
            bool check = false, adapt = false;
            if (s.check_termination != 0 && --until_check == 0) { check = true; until_check = s.check_termination; }
            if (s.adaptive_rho && --until_adapt == 0) { adapt = true; until_adapt = s.adaptive_rho_interval; }
            if (check || adapt) {   // residuals_update, box_admm.hpp:398-415
                const long long r0 = dbg ? clock64() : 0;
#pragma unroll
                for (int e = 0; e < SLOTS; ++e) *xst[e] = xv[e];
                *dst = ya;
                lds_order();
                int zr = 0; asm volatile("" : "+v"(zr));   // (as in the factorisation: keeps the block reads of this branch inside it)
                const double* hbR = hblk + zr; const double* jbR = jblk + zr;
                const JView<Model, NNODES> jv{Dm, nsr, jbR, jbR, PP};
                const double Ax = jv.rowdot(ci, xsc);
                double nrmP = 0.0, rq = 0.0, rd = 0.0, nx_ = 0.0;
                double pcur = 0.0, hxp = 0.0;   // NP = 1: the parameter, and row p of H x — the dense chain over all columns ascending (every lane forms it)
                if constexpr (NPAR > 0) { const double* hbB = hbrd + zr; pcur = xsc[N0]; hxp = seq_dot(hbB, xsc, N); }
#pragma unroll
                for (int e = 0; e < SLOTS; ++e) {
                    const Role r = role(e);
                    double hb[D], xn[D];   // row g of H: the node's block, columns ascending (x columns, then u columns)
#pragma unroll
                    for (int c = 0; c < D; ++c) hb[c] = hbR[r.k * DD + c * D + r.c];
#pragma unroll
                    for (int c = 0; c < NX; ++c) xn[c] = xo[e][c];
#pragma unroll
                    for (int c = 0; c < NU; ++c) xn[NX + c] = uo[e][c];
                    double Hx = 0.0;
#pragma unroll
                    for (int c = 0; c < D; ++c) Hx += hb[c] * xn[c];
                    if constexpr (NPAR > 0) { const double hc = hbrd[zr + N + (r.isp ? 0 : r.g)]; Hx += hc * pcur; Hx = r.isp ? hxp : Hx; }   // the border column is the last product of the row's chain
                    const double aty = jv.coldot(r.g, dsc, r.pv);
                    const double np_ = fmax(fmax(fabs(Hx), fabs(aty)), fmax(fabs(hv[e]), fabs(yb[e])));
                    nrmP = r.pv ? fmax(nrmP, np_) : nrmP;
                    rq = r.pv ? fmax(rq, fabs(xv[e] - qv[e])) : rq;
                    rd = r.pv ? fmax(rd, fabs(((Hx + hv[e]) + aty) + yb[e])) : rd;
                    nx_ = r.pv ? fmax(nx_, fabs(xv[e])) : nx_;
                }
                max_Ax_z_norm = wave_max(fmax(isC ? fmax(fabs(Ax), fabs(zv)) : 0.0, nx_));
                max_Hx_ATy_h_norm = wave_max(nrmP);
                const double rp = wave_max(isC ? fabs(Ax - zv) : 0.0);
                res_prim = rp + wave_max(rq);
                res_dual = wave_max(rd);
                lds_order();
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
                    for (int e = 0; e < SLOTS; ++e) {   // update_kkt_rho, box_admm.hpp:448-452
                        const Role r = role(e);
                        const double prev = rhob[e];
                        rhob[e] = rho_of(typ[e], rho); rhobinv[e] = 1.0 / rhob[e];
                        kdv[e] = kdv[e] + (rhob[e] - prev);
                    }
                    rhoA = rho_of(typA, rho); rinvA = 1.0 / rhoA;
                    ++rho_updates;
                    refactor = true;
                    ++iter;
                    lds_order();
                    break;
                }
            }
        }
        if (!refactor) running = false;
    }
    const bool gave_up = status == GAVE_UP;
    if (gave_up) status = PMPC_QP_UNSOLVED; else
    if (iter > s.max_iter) status = PMPC_QP_MAX_ITER_EXCEEDED;
    bool nf = false;
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) {
        const Role r = role(e);
        if (r.pv) { out_x[r.g] = xv[e]; out_y[M + r.g] = yb[e]; }
        nf = nf || (r.pv && ((xv[e] - xv[e]) + (yb[e] - yb[e])) != 0.0);
    }
    if (isC) out_y[ci] = ya;
    nf = nf || (isC && (ya - ya) != 0.0);
    const bool bad = __builtin_amdgcn_ballot_w64(nf) != 0;
    info.status = status; info.iter = iter; info.rho_updates = rho_updates; info.flags = (bad ? PMPC_FLAG_NONFINITE : 0) | (gave_up ? PMPC_FLAG_ILLCOND : 0);
    info.rho_estimate = rho_estimate; info.res_prim = res_prim; info.res_dual = res_dual;
}

}  // namespace pmpc
