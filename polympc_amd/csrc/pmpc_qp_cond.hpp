// polympc_amd — CONDENSED register-resident box-ADMM QP solve for the QPs of the fused SQP kernel with 65..128 KKT rows, at most 112 variables and at most
// 64 constraint rows (one wavefront per QP; config B: n = 66, m = 44; the reference's 16-node robot grid: n = 80, m = 48; its 11-node grid: n = 55, m = 33).
//
// The constraint block of boxADMM's KKT matrix (box_admm.hpp:209-223) is diagonal, -1/rho, so the constraint rows are eliminated in closed form — what
// the constraint-first sweep of pmpc_qp_reg.hpp does inside the full inverse — and are then NOT CARRIED AT ALL: only
//     S = H + sigma I + rho_box + A' diag(rho) A          (n x n instead of (n + m) x (n + m))
// is inverted, W = -S^{-1} by the blocked sweep of pmpc_qp_reg2.hpp / pmpc_qp_reg.hpp in 16 x 16 fp64 accumulator tiles (config B: 17 block steps on 15 stored tiles
// instead of 28 on 28, 25 operand tiles of the mat-vec instead of 49), and every ADMM iteration solves
//     t = r1 + A'(rho o r2),      x = S^{-1} t,      nu = rho o (A x - r2)
// with the two products formed from the per-node blocks of A and two small tables of the differentiation matrix in LDS (fma chains: the D~ entries of
// the column / row over the nodes ascending, then the own node's block — no HBM traffic, no selects). A' diag(rho) A is a rank-m update on the matrix cores between the staging of H and the sweep
// (RegKkt2::rank_update: operands rho_j A(j, .) and A(j, .) for the constraint rows j in groups of four, the k-ascending fma chain of the MFMA).
// Same pivots as the constraint-first sweep; measured on the QP streams of configs B / R against the reference's pivoted LDL^T: every QP keeps its
// ADMM iteration count, max |d res| 2.9e-10 (the two-rows-per-lane full inverse: 1.1e-9). CPU restatement of exactly this order: PIVOT_CONDSWEEP
// (the test suite's checker) — the kernels are checked against it bit for bit.
// The QP entry points without structure information (pmpc_qp_solve_batch) stay on the full two-rows-per-lane inverse: the products with a dense A
// would cost more than the rows they save.
// Round 6: one parameter (NP = 1: its dense column of A as a wave reduction), path-constraint rows (NG > 0: own-node blocks without a D~ row), H x of the dual residual from the KKT
// identity of the last solve, and — hook builds (WS) — the D~ tables per state index and the node blocks read back from a Ruiz-scaled workspace.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_qp_reg2.hpp"
#include "pmpc_jview.hpp"

namespace pmpc {

#ifndef PMPC_COND_NV
#define PMPC_COND_NV 25
#endif
// the swept inverse of S: the two-rows-per-lane tile set with PMPC_COND_NV of its operand tiles in arch VGPRs (25 tiles at 66..80 rows: the rest lives
// in the accumulation file and costs two v_accvgpr_read per entry and ADMM iteration)
template <int NN, bool SMALL = (NN <= WAVE)> struct CondKktSel { using type = RegKkt2<NN, PMPC_COND_NV>; };
// at most 64 variables — grids of 65..128 KKT rows whose primal block fits one row per lane: the one-row-per-lane tile set of pmpc_qp_reg.hpp (16 tiles
// = 128 registers, a register-only DPP mat-vec); one wavefront per SIMD like the large variant (PMPC_COND1_WAVES, pmpc_launch.hpp: measured)
template <int NN> struct CondKktSel<NN, true> { using type = RegKkt<NN>; };
template <int NN> using CondKkt = typename CondKktSel<NN>::type;

// LDS the solver needs beside the staging of CondKkt<NN>: nothing — the exchange vectors alias the staging (free between factorisations)
template <int NN, int MM>
struct CondDims {
    static_assert(NN > 0 && NN <= 128 && MM > 0 && MM <= WAVE, "condensed register QP: at most 128 variables and 64 constraint rows");
    static constexpr int N = NN + MM;
    static constexpr bool SMALL = NN <= WAVE;
    static constexpr int SL = SMALL ? 1 : 2;                      // primal slots per lane
    static constexpr int RHS_OFF = []() constexpr { if constexpr (NN <= WAVE) return 0; else return (int)RegKkt2<NN, PMPC_COND_NV>::RHS_OFF; }();
    static constexpr int US_OFF = RHS_OFF + 128;                  // rho o r2 / y (MM entries) behind the rhs exchange buffer of RegKkt2::apply
    static constexpr int XS_OFF = US_OFF + 64;                    // x (NN entries)
    static constexpr int PB_OFF = XS_OFF + 128;                   // residual evaluation: products of the few primal rows of the second slot
    static constexpr int TAB_OFF = PB_OFF + (SMALL ? 0 : 4 * NN); // D~ tables of the sparse products (cond_build_tables), rebuilt at every QP
    template <int NNODES> static constexpr int nnp() { return lds_row_stride(NNODES); }   // bank-conflict-free row stride (pmpc_jview.hpp)
    template <int NNODES> static constexpr int tab_doubles() { return (4 * NNODES + 1) * nnp<NNODES>(); }
    // WS (the hook builds, round 6): one set of four tables PER STATE INDEX, filled from the workspace (a Ruiz-scaled A(r, c) = E_r D~ D_c is no longer a function of the two nodes alone)
    template <int NNODES, int NX> static constexpr int tab_doubles_ws() { return (4 * NNODES * NX + 1) * nnp<NNODES>(); }
};

// D~ as two dense tables in LDS: Dt[r NNP + k] = D~(r, k) — the differentiation-matrix entry of equality row node r on the state columns of node k
// (continuous_ocp.hpp:817-827, :845-846), 0 on the own node (that entry lives in the node block) and outside the row's segment — and its transpose
// DtT[k NNP + r]; then the parts of both BELOW the own node (Dlo[r][k] = D~(r, k) for k < r, DtTlo[k][r] = D~(r, k) for r < k, 0 elsewhere: the residual
// evaluation walks a row / column in the reference's ascending order — entries before the own node's block, the block, entries behind it — and forms
// the part behind as table - lower part, exactly); one all-zero row at the end (read by control columns). Run-time P: these kernels are compiled per
// node count.
template <int NNODES>
__device__ __forceinline__ void cond_build_tables(const double* Dm, int P, double* Dt) {
    constexpr int NNP = lds_row_stride(NNODES);
    double* DtT = Dt + NNODES * NNP;
    double* Dlo = DtT + NNODES * NNP;
    double* DtTlo = Dlo + NNODES * NNP;
    for (int e = lane_id(); e < (4 * NNODES + 1) * NNP; e += WAVE) Dt[e] = 0.0;
    lds_order();
    const int P1 = P + 1;
    for (int e = lane_id(); e < NNODES * NNODES; e += WAVE) {
        const int r = e / NNODES, k = e - r * NNODES;
        const bool lastr = r == NNODES - 1;
        const int kbr = lastr ? NNODES - 1 - P : (r / P) * P;
        const int rowr = lastr ? P : r - kbr;
        const int t = k - kbr;
        const bool cpl = k != r && (unsigned)t <= (unsigned)P;
        const double dv = Dm[lastr ? P1 * P1 + (cpl ? t : 0) : rowr + (cpl ? t : 0) * P1];
        const double v = cpl ? dv : 0.0;
        Dt[r * NNP + k] = v;
        DtT[k * NNP + r] = v;
        Dlo[r * NNP + k] = (k < r) ? v : 0.0;
        DtTlo[k * NNP + r] = (r < k) ? v : 0.0;
    }
    lds_order();
}

// The same tables from the WORKSPACE, one set per state index q (set q at q * 4 * NNODES * NNP; one all-zero row behind the last set): entry (r, k) of set q is
// A((r, q), (k, q)) as it stands in the stacked workspace — the Ruiz preconditioner (qp_preconditioners.hpp:114-220) has rescaled it by the row's and the column's factor,
// which differ from state to state. Without scaling the entries ARE D~(r, k): the hook builds of the condensed kernel use this form whatever the policy.
// Hs: the stacked workspace [H ; A] (leading dimension N = NN + MM), A(r, c) at Hs[c N + NN + r].
template <int NNODES, int NX, int NN, int MM>
__device__ __forceinline__ void cond_build_tables_ws(const double* __restrict__ Hs, int P, double* Dt) {
    constexpr int NNP = lds_row_stride(NNODES), SET = 4 * NNODES * NNP, N = NN + MM;
    for (int e = lane_id(); e < NX * SET + NNP; e += WAVE) Dt[e] = 0.0;
    lds_order();
    for (int e = lane_id(); e < NX * NNODES * NNODES; e += WAVE) {
        const int q = e / (NNODES * NNODES), rk = e - q * (NNODES * NNODES);
        const int r = rk / NNODES, k = rk - r * NNODES;
        const bool lastr = r == NNODES - 1;
        const int kbr = lastr ? NNODES - 1 - P : (r / P) * P;
        const bool cpl = k != r && (unsigned)(k - kbr) <= (unsigned)P;
        const double av = Hs[(size_t)(k * NX + q) * N + NN + (r * NX + q)];
        const double v = cpl ? av : 0.0;
        double* T = Dt + q * SET;
        T[r * NNP + k] = v;
        T[NNODES * NNP + k * NNP + r] = v;
        T[2 * NNODES * NNP + r * NNP + k] = (k < r) ? v : 0.0;
        T[3 * NNODES * NNP + k * NNP + r] = (r < k) ? v : 0.0;
    }
    lds_order();
}

// boxADMM::solve_impl (7-argument form: zero guesses, box_admm.hpp:81-86). H: the stacked workspace [H ; A] of the fused SQP kernel ((NN + MM) x NN,
// column-major, leading dimension NN + MM; the LOWER triangle of H is read for S, as Eigen::LDLT does; the full rows for H x); h / bounds: LDS vectors;
// tr: CondKkt<NN>::TRI doubles of LDS; jv: the block-sparse view of A.
template <int NN, int MM, class JV, bool WS = false>   // WS: a hook build — with ws_on (wave-uniform; the Ruiz preconditioner has rescaled the workspace, the caller has refreshed the node blocks from it and sized the staging for it) the D~ tables are built per state index from the workspace
__device__ __forceinline__ void boxadmm_solve_cond(const double* __restrict__ H, const double* h, const double* Alb, const double* Aub, const double* xlb,
                                                   const double* xub, const pmpc_qp_settings& s, pmpc_qp_info& info, double* out_x, double* out_y, double* tr,
                                                   const JV& jv, long long* dbg = nullptr, long long* tm = nullptr, bool ws_on = false) {
    using CD = CondDims<NN, MM>;
    const bool ws = WS && __builtin_amdgcn_readfirstlane((int)ws_on) != 0;
    constexpr int N = CD::N, SL = CD::SL;
    constexpr bool SMALL = CD::SMALL;
    const int ln = lane_id();
    // primal slot e: variable lane + 64 e; constraint row: lane (lanes [0, MM))
    bool isP[2] = {false, false}; int lp[2] = {0, 0};
    double hv[2] = {0.0, 0.0}, lo[2] = {0.0, 0.0}, hi[2] = {0.0, 0.0}; int typ[2] = {0, 0};
#pragma unroll
    for (int e = 0; e < SL; ++e) {
        const int idx = ln + 64 * e;
        isP[e] = idx < NN; lp[e] = isP[e] ? idx : 0;
        hv[e] = isP[e] ? h[lp[e]] : 0.0;
        lo[e] = isP[e] ? xlb[lp[e]] : 0.0; hi[e] = isP[e] ? xub[lp[e]] : 0.0;
        typ[e] = classify_bounds(lo[e], hi[e]);
    }
    const bool isC = ln < MM;
    const int rc = isC ? ln : 0;
    const double clo = isC ? Alb[rc] : 0.0, chi = isC ? Aub[rc] : 0.0;
    const int ctyp = classify_bounds(clo, chi);
    auto lane_near = [](int zo) -> unsigned { unsigned l; asm("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l) : "v"(zo)); return l; };
    // H(max(i, j), min(i, j)) for row i = lane + 64 e (0.0 on lanes without such a row), j < NN
    auto Hlow = [&](int j, int e, int zo) -> double {
        const unsigned i = lane_near(zo) + 64u * (unsigned)e;
        const bool live = i < (unsigned)NN;
        const unsigned ic = live ? i : 0u;
        unsigned b = (ic < (unsigned)j) ? ((unsigned)j + ic * (unsigned)N) : (ic + (unsigned)(j * N));
        b += (unsigned)zo; asm("" : "+v"(b));
        const double v = H[b];
        return live ? v : 0.0;
    };
    // A(k, i), i = lane + 64 e (0.0 on lanes without such a variable)
    auto Acol = [&](int k, int e, int zo) -> double {
        const unsigned i = lane_near(zo) + 64u * (unsigned)e;
        const bool prim = i < (unsigned)NN;
        unsigned b = (prim ? i : 0u) * (unsigned)N + (unsigned)NN + (unsigned)zo; asm("" : "+v"(b));
        const double v = H[b + (unsigned)k];
        return prim ? v : 0.0;
    };
    auto xbc = [&](const double (&v)[2], int j) -> double { return (j < 64) ? bcast_lane(v[0], j & 63) : bcast_lane(v[1], (j - 64) & 63); };

    // state: xv / qv / yb on the primal slots, zv / ya on the constraint lanes
    double xv[2] = {0.0, 0.0}, qv[2] = {0.0, 0.0}, yb[2] = {0.0, 0.0}, zv = 0.0, ya = 0.0;
    double rho = s.rho;
    int rho_updates = 1;
    double rhob[2] = {1.0, 1.0}, rhobinv[2] = {1.0, 1.0}, kd[2] = {0.0, 0.0};
#pragma unroll
    for (int e = 0; e < SL; ++e) {
        rhob[e] = rho_of(typ[e], rho);
        rhobinv[e] = 1.0 / rhob[e];
        double d = H[(size_t)lp[e] * N + lp[e]]; d += s.sigma; d += rhob[e];
        kd[e] = isP[e] ? d : 0.0;   // (rows >= NN: padding, exact zeros, never swept)
    }
    double rhoc = rho_of(ctyp, rho), rhocinv = 1.0 / rhoc;

    CondKkt<NN> K;
    double* us = tr + CD::US_OFF; double* xs = tr + CD::XS_OFF;
    // the two sparse products of an ADMM iteration as fma chains whose coefficients come from LDS at per-lane bases with immediate offsets — no selects,
    // no index arithmetic in the loop (the multiply-add products of pmpc_jview.hpp, built for the residuals' reference order, cost 900 of the 1500
    // instructions of an iteration here):
    //   column c of node jn, state qx:  t = r1;  t = fma(D~(k, jn), u(k, qx), t) for the nodes k ascending (0 on the own node and outside the segments
    //                                   that hold jn; a control column reads the all-zero row);  t = fma(J((jn, q), c), u(jn, q), t) for q ascending
    //   row (k, q):                     a = 0;   a = fma(D~(k, j), x(j, q), a) for the nodes j ascending;  then the own node's block, columns ascending
    constexpr int NX = JV::NX, NU = JV::NU, NPAR = JV::NP, NG = JV::NG, NDER = JV::NDER, JBS = JV::JBS, NNODES = MM / (NX + NG), NNP = lds_row_stride(NNODES), VARX = NX * NNODES;
    constexpr int P0 = (NX + NU) * NNODES;   // NP = 1 (round 6): the parameter is the last primal variable, its column of A is DENSE (rows ascending: entry NX + NU of every row's block)
    constexpr int ME = NX * NNODES;          // NG > 0 (round 6): the path-constraint rows ME + k NG + g follow the equality rows (continuous_ocp.hpp:546-575); such a row holds its own node's block only
    constexpr int NGC = NG > 0 ? NG : 1;
    static_assert(NPAR <= 1 && NNODES * (NX + NG) == MM && P0 + NPAR == NN, "condensed register QP: at most one parameter");
    double* Dt = tr + CD::TAB_OFF;
    static_assert(CD::TAB_OFF + CD::template tab_doubles<NNODES>() <= CondKkt<NN>::TRI, "tables fit the staging");   // (ws: the launcher sizes the staging, cond_qp_staging_ws)
    constexpr int SET = 4 * NNODES * NNP;                      // one set of four tables
    const double* ZROW = Dt + (ws ? NX : 1) * SET;             // the all-zero row
    const double* DtT = Dt + NNODES * NNP;
    const double *cD[2] = {nullptr, nullptr}, *cU[2] = {nullptr, nullptr}, *cB[2] = {nullptr, nullptr}, *cV[2] = {nullptr, nullptr};   // per primal slot: D~ column, u at the column's state index, own-node block column, u of the own node
    const double *cG[2] = {nullptr, nullptr}, *cW[2] = {nullptr, nullptr};   // NG > 0: the column inside its own node's path-constraint rows, u of those rows
    bool isp[2] = {false, false};   // this slot holds the parameter (NP = 1): its entry of A' u is a wave reduction, not a chain over the tables
    // The per-lane index arithmetic below is written WITHOUT lane-divergent control flow (both quotients are formed, a bit mask selects): a ternary around an integer division
    // becomes an if / else over the lanes, and a divergent block in a kernel at the register limit is where hipcc 7.2 has twice placed the copy of a live-range split that then
    // keeps only the active lanes' values (DESIGN.md hazard 3: round 5 in the column setup, late round 6 in the row setup of the first NG > 0 hook build)
    auto isel = [](bool cnd, int a, int b) -> int { const int mk = -(int)cnd; return (a & mk) | (b & ~mk); };
#pragma unroll
    for (int e = 0; e < SL; ++e) {
        const int c = lp[e];
        const bool xcol = c < VARX;
        isp[e] = NPAR > 0 && isP[e] && c >= P0;
        const int cu = isel(isp[e], 0, c - VARX);     // (the parameter's lane walks a control column's addresses: its chain is discarded)
        const int cuc = cu < 0 ? 0 : cu;              // (a state column: any valid control index, discarded by the select)
        const int jnx = c / NX, jnu = cuc / NU;
        const int jn = isel(xcol, jnx, jnu);
        const int dcol = isel(xcol, c - jnx * NX, NX + (cuc - jnu * NU));
        const int doff = isel(xcol, (ws ? dcol * SET : 0) + jn * NNP + NNODES * NNP, (ws ? NX : 1) * SET);   // offset from Dt: the column's row of the transposed table / the all-zero row
        cD[e] = Dt + doff;
        cU[e] = us + isel(xcol, dcol, 0);
        cB[e] = jv.jblk + (jn * NX) * JBS + dcol;
        cV[e] = us + jn * NX;
        cG[e] = jv.jblk + (ME + jn * NG) * JBS + dcol;   // (gblk = jblk + ME JBS: the launcher carves them as one array, pmpc_launch.hpp)
        cW[e] = us + ME + jn * NG;
    }
    const bool req = NG == 0 || rc < ME;           // equality row (node rk, state rq) or path-constraint row (node rk): the latter reads the all-zero row of the D~ tables
    const int rcg = rc < ME ? 0 : rc - ME;         // (index among the path rows; an equality row: 0, discarded by the select)
    const int rke = rc / NX, rkg = rcg / NGC;
    const int rk = isel(req, rke, rkg), rq = isel(req, rc - rke * NX, 0);
    const double* rD = Dt + isel(req, (ws ? rq * SET : 0) + rk * NNP, (ws ? NX : 1) * SET);   // constraint row: D~ row (a path row: the all-zero row), x at the row's state index, own-node block row, x / u of the own node
    const double* rX = xs + rq;
    const double* rB = jv.jblk + rc * JBS;
    const double* rV = xs + rk * NX;
    const double* rW = xs + VARX + rk * NU;
    // NP = 1: A(r, p) of this lane's row — constant over the QP (one LDS read), 0 on the lanes without a row
    double arp = 0.0;
    if constexpr (NPAR > 0) { const double v_ = jv.jblk[rc * JBS + NX + NU]; arp = isC ? v_ : 0.0; }
    (void)arp;
    constexpr bool SLOT1_STATES = VARX > 64;       // state columns in the second slot?
    constexpr int CH = SMALL ? 4 : NNODES;         // D~ entries per batch of LDS reads in the two products
    auto coldot_fma = [&](int e, double init) -> double {
        double a = init;
        if (e == 0 || SLOT1_STATES) {
#pragma unroll
            for (int k0 = 0; k0 < NNODES; k0 += CH) {   // (CH entries per batch of LDS reads: the 256-register build has no room for all of them at once)
                double dv[CH], uv[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) { const int kk = (k0 + k < NNODES) ? k0 + k : 0; dv[k] = cD[e][kk]; uv[k] = cU[e][kk * NX]; }
#pragma unroll
                for (int k = 0; k < CH; ++k) if (k0 + k < NNODES) a = fma(dv[k], uv[k], a);
                if constexpr (CH < NNODES) sched_fence();
            }
        }
        double bv[NX], vv[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) { bv[q] = cB[e][q * JBS]; vv[q] = cV[e][q]; }
#pragma unroll
        for (int q = 0; q < NX; ++q) a = fma(bv[q], vv[q], a);
        if constexpr (NG > 0) {   // the own node's path-constraint rows, behind the equality rows
            double gv[NGC], wv[NGC];
#pragma unroll
            for (int g = 0; g < NG; ++g) { gv[g] = cG[e][g * JBS]; wv[g] = cW[e][g]; }
#pragma unroll
            for (int g = 0; g < NG; ++g) a = fma(gv[g], wv[g], a);
        }
        return a;
    };
    auto rowdot_fma = [&]() -> double {
        double a = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < NNODES; j0 += CH) {
            double dv[CH], xq[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) { const int jj = (j0 + j < NNODES) ? j0 + j : 0; dv[j] = rD[jj]; xq[j] = rX[jj * NX]; }
#pragma unroll
            for (int j = 0; j < CH; ++j) if (j0 + j < NNODES) a = fma(dv[j], xq[j], a);
            if constexpr (CH < NNODES) sched_fence();
        }
        double bv[NDER], xb[NDER];
#pragma unroll
        for (int i = 0; i < NX; ++i) { bv[i] = rB[i]; xb[i] = rV[i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) { bv[NX + i] = rB[NX + i]; xb[NX + i] = rW[i]; }
#pragma unroll
        for (int i = 0; i < NX + NU; ++i) a = fma(bv[i], xb[i], a);   // (NP = 1: the caller appends the parameter's term)
        return a;
    };
    int status = PMPC_QP_UNSOLVED;
    const double alpha = s.alpha;
    double max_Ax_z_norm = 0.0, max_Hx_ATy_h_norm = 0.0, res_prim = 1.0, res_dual = 1.0, rho_estimate = 0.0;
    double r1l[2] = {0.0, 0.0}, nul = 0.0;   // first right-hand side and multipliers nu of the LAST solve (the residual evaluation's H x, see there)
    int iter = 1;
    int until_check = s.check_termination, until_adapt = s.adaptive_rho_interval;
    bool running = true;
    while (running) {
        {   // construct_kkt_matrix + factorise_kkt_matrix (box_admm.hpp:209-223, :336-341) in condensed form
            const long long f0 = dbg ? clock64() : 0;
            const double rc_now = rhoc;
            if constexpr (SMALL) {
                K.invert(ln, tr, kd[0], [&](int j, int z) -> double { return Hlow(j < NN ? j : 0, 0, z); }, tm, 0.0,
                         [&](typename CondKkt<NN>::d4 (&T)[CondKkt<NN>::NT][CondKkt<NN>::NT], double* PA, double* PB, int l, int lr, int lc) {
                             CondKkt<NN>::template rank_update<MM>(T, l, lr, lc, PA, PB, [&](int j, int z) -> double { return Acol(j, 0, z); },
                                                                   [&](int j) -> double { return bcast_lane(rc_now, j); });
                         });
            } else {
                K.invert(ln, tr, kd[0], kd[1], [&](int j, int e, int z) -> double { return Hlow(j < NN ? j : 0, e, z); }, tm,
                         [&](CondKkt<NN>& Kr, double* PA, double* PB, int l, int lr, int lc) {
                             Kr.template rank_update<MM>(l, lr, lc, PA, PB, [&](int j, int e, int z) -> double { return Acol(j, e, z); },
                                                         [&](int j) -> double { return bcast_lane(rc_now, j); });
                         });
            }
            if (ws) { if constexpr (WS) cond_build_tables_ws<NNODES, NX, NN, MM>(H, jv.P, Dt); }
            else cond_build_tables<NNODES>(jv.D, jv.P, Dt);   // (the staging they live in was the sweep's)
            if (dbg) dbg[0] += clock64() - f0;
        }
        bool refactor = false;
        while (iter <= s.max_iter) {
            int nrun = s.max_iter - iter + 1;
            if (s.check_termination != 0 && until_check < nrun) nrun = until_check;
            if (s.adaptive_rho && until_adapt < nrun) nrun = until_adapt;
            for (int kk = 0; kk < nrun; ++kk) {
                const double zprev = zv;
                const double r2 = zv - rhocinv * ya;                       // compute_kkt_rhs, box_admm.hpp:351-355
                const double uval = rhoc * r2;
                if (isC) us[rc] = uval;
                lds_order();
                double psum = 0.0;   // NP = 1: sum_r A(r, p) u_r — lane r forms its product, the 64 products are added by the DPP tree of wave_sum (restated: cond_wave_dot)
                if constexpr (NPAR > 0) psum = wave_sum(arp * uval);
                double t[2] = {0.0, 0.0}, sol[2] = {0.0, 0.0};
#pragma unroll
                for (int e = 0; e < SL; ++e) {
                    // (two wavefronts per SIMD, 256 registers: h and the bounds are re-read from their LDS vectors — kept in registers they were spilled, and
                    //  six scratch reloads per ADMM iteration cost more than six LDS reads)
                    const double hve = SMALL ? (isP[e] ? h[lp[e]] : 0.0) : hv[e];
                    const double rhs1 = ((s.sigma * xv[e] - hve) + rhob[e] * qv[e]) - yb[e];
                    const double a = coldot_fma(e, rhs1);
                    t[e] = isP[e] ? a : 0.0;
                    if constexpr (NPAR > 0) t[e] = isp[e] ? rhs1 + psum : t[e];
                    r1l[e] = rhs1;
                }
                lds_order();
                if constexpr (SMALL) sol[0] = K.apply(t[0]);
                else K.apply(t[0], t[1], tr, ln, sol[0], sol[1]);
#pragma unroll
                for (int e = 0; e < SL; ++e) if (isP[e]) xs[lp[e]] = sol[e];
                lds_order();
                double ax = rowdot_fma();
                if constexpr (NPAR > 0) ax = fma(arp, xs[P0], ax);         // the parameter: the last column of the row
                const double nu = rhoc * (ax - r2);
                nul = nu;
                lds_order();
                {
                    const double zt = zprev + rhocinv * (nu - ya);
                    double zz = alpha * zt;
                    zz += (1 - alpha) * zprev + rhocinv * ya;
                    const double cl_ = SMALL ? (isC ? Alb[rc] : 0.0) : clo, ch_ = SMALL ? (isC ? Aub[rc] : 0.0) : chi;
                    zz = fmin(fmax(zz, cl_), ch_);
                    const double yC = ya + rhoc * ((alpha * zt + (1 - alpha) * zprev) - zz);
                    zv = isC ? zz : 0.0; ya = isC ? yC : 0.0;
                }
#pragma unroll
                for (int e = 0; e < SL; ++e) {
                    double xx = alpha * sol[e];
                    xx += (1 - alpha) * xx;  // quirk Q1
                    double qq = xx + rhobinv[e] * yb[e];
                    const double lo_ = SMALL ? (isP[e] ? xlb[lp[e]] : 0.0) : lo[e], hi_ = SMALL ? (isP[e] ? xub[lp[e]] : 0.0) : hi[e];
                    qq = fmin(fmax(qq, lo_), hi_);
                    const double yP = yb[e] + rhob[e] * (xx - qq);
                    xv[e] = isP[e] ? xx : 0.0; qv[e] = isP[e] ? qq : 0.0; yb[e] = isP[e] ? yP : 0.0;
                }
            }
            iter += nrun - 1;   // the last iteration performed
            bool check = false, adapt = false;
            if (s.check_termination != 0) { until_check -= nrun; if (until_check == 0) { check = true; until_check = s.check_termination; } }
            if (s.adaptive_rho) { until_adapt -= nrun; if (until_adapt == 0) { adapt = true; until_adapt = s.adaptive_rho_interval; } }
            if (check || adapt) {  // residuals_update, box_admm.hpp:398-415: one add chain per row, columns ascending
                const long long r0 = dbg ? clock64() : 0;
                int zr = 0;
                asm volatile("" : "+v"(zr));
                const double probe = ((xv[0] - xv[0]) + (xv[1] - xv[1])) + (ya - ya);
                const bool finite = __builtin_amdgcn_ballot_w64(probe != 0.0) == 0;   // (a non-finite iterate takes the dense loops: 0 * inf = NaN on the structural zeros of A)
                const int sl = (int)lane_near(zr);
                double acc[2] = {0.0, 0.0}, aty[2] = {0.0, 0.0}, axz = 0.0;
                constexpr int NP1 = SMALL ? 0 : NN - 64;   // primal rows of the second slot
                constexpr bool FEW1 = !SMALL && NP1 <= 4;   // few of them: products through LDS; otherwise their lanes load their rows
                constexpr int RCS = 22;
                // H x of the dual residual (qp_base.hpp:240-252) WITHOUT re-reading H (round 6): the solve that produced x satisfies
                //     (H + sigma I + rho_box) x~ + A' nu = r1      =>      H x~ = (r1 - A' nu) - (sigma + rho_box) o x~ ,
                // and x = x~ for alpha = 1 (quirk Q1: x = alpha (2 - alpha) x~). r1 and nu are the last solve's, A' nu is one more fma chain over the tables
                // (the first product of the solve, started from 0). The reference forms H x by a second mat-vec with H; the two differ by the linear solve's own
                // residual (~1e-13 relative). Measured with the CPU restatement (PIVOT_CONDSWEEP restates this; EXPERIMENTS.md round 6) on the streams of configs
                // A / D / B / R: NO instance changes its SQP or ADMM iteration counts, max |dx| 2e-10 (A) .. 2e-7 absolute on a control bounded by 9000 (B).
                // What it buys: the 35 KB of H (config B) were re-read from L2 / HBM at each of the ~6 residual evaluations per QP — 193 KB fetched per QP against
                // 62 KB algorithmic — behind 66 + dependent broadcast / multiply / add chains. alpha != 1 or a non-finite iterate: the mat-vec with H, as before.
                const bool ident = finite && __builtin_amdgcn_readfirstlane((int)(alpha == 1.0)) != 0;
                if (ident) {
                    if (isC) us[rc] = nul;
                    lds_order();
                    double pnu = 0.0;
                    if constexpr (NPAR > 0) pnu = wave_sum(arp * nul);
#pragma unroll
                    for (int e = 0; e < SL; ++e) {
                        double atnu = coldot_fma(e, 0.0);
                        if constexpr (NPAR > 0) atnu = isp[e] ? pnu : atnu;
                        double hx = r1l[e] - atnu;
                        hx -= (s.sigma + rhob[e]) * xv[e];
                        acc[e] = isP[e] ? hx : 0.0;
                    }
                    lds_order();
                }
                if (finite) {
#pragma unroll
                    for (int e = 0; e < SL; ++e) if (isP[e]) xs[lp[e]] = xv[e];
                    if (isC) us[rc] = ya;
                    lds_order();
                    // A' y and A x in the reference's order (multiply, then add; ascending index: the entries before the own node's block, the block, the
                    // entries behind it — pmpc_jview.hpp states why these are the dense chains bit for bit) from the same tables as the fma products
                    const double* DtTlo = DtT + 2 * NNODES * NNP;
                    const double* Dlo = DtT + NNODES * NNP;
#pragma unroll
                    for (int e = 0; e < SL; ++e) {
                        double a = 0.0;
                        const bool dpart = (e == 0 || SLOT1_STATES);
                        double dv[NNODES], lv[NNODES], uv[NNODES], bv[NX], vv[NX];
                        if (dpart) {
                            const double* cl = cD[e] + ((cD[e] != ZROW) ? 2 * NNODES * NNP : 0);   // (a control column keeps the all-zero row)
#pragma unroll
                            for (int k = 0; k < NNODES; ++k) { dv[k] = cD[e][k]; lv[k] = cl[k]; uv[k] = cU[e][k * NX]; }
                        }
#pragma unroll
                        for (int q = 0; q < NX; ++q) { bv[q] = cB[e][q * JBS]; vv[q] = cV[e][q]; }
                        if (dpart) {
#pragma unroll
                            for (int k = 0; k < NNODES; ++k) a += lv[k] * uv[k];
                        }
#pragma unroll
                        for (int q = 0; q < NX; ++q) a += bv[q] * vv[q];
                        if (dpart) {
#pragma unroll
                            for (int k = 0; k < NNODES; ++k) a += (dv[k] - lv[k]) * uv[k];
                        }
                        if constexpr (NG > 0) {   // rows ME + jn NG + g: the last rows of the column
                            double gv[NGC], wv[NGC];
#pragma unroll
                            for (int g = 0; g < NG; ++g) { gv[g] = cG[e][g * JBS]; wv[g] = cW[e][g]; }
#pragma unroll
                            for (int g = 0; g < NG; ++g) a += gv[g] * wv[g];
                        }
                        aty[e] = isP[e] ? a : 0.0;
                    }
                    if constexpr (NPAR > 0) {   // the parameter's column in the reference's order: sum_r A(r, p) y_r, rows ascending (multiply, then add) — a serial chain every lane walks, the parameter's lane keeps it
                        double a = 0.0;
                        constexpr int CHP = 11;
#pragma unroll
                        for (int r0 = 0; r0 < MM; r0 += CHP) {
                            double av[CHP], yv[CHP];
#pragma unroll
                            for (int r = 0; r < CHP; ++r) { const int rr = (r0 + r < MM) ? r0 + r : 0; av[r] = jv.jblk[rr * JBS + NX + NU]; yv[r] = us[rr]; }
#pragma unroll
                            for (int r = 0; r < CHP; ++r) if (r0 + r < MM) a += av[r] * yv[r];
                        }
#pragma unroll
                        for (int e = 0; e < SL; ++e) aty[e] = isp[e] ? a : aty[e];
                    }
                    {
                        double a = 0.0;
                        const double* rl = req ? rD + 2 * NNODES * NNP : ZROW;
                        double dv[NNODES], lv[NNODES], xq[NNODES], bv[NDER], xb[NDER];
#pragma unroll
                        for (int j = 0; j < NNODES; ++j) { dv[j] = rD[j]; lv[j] = rl[j]; xq[j] = rX[j * NX]; }
#pragma unroll
                        for (int i = 0; i < NX; ++i) { bv[i] = rB[i]; xb[i] = rV[i]; }
#pragma unroll
                        for (int i = 0; i < NU; ++i) { bv[NX + i] = rB[NX + i]; xb[NX + i] = rW[i]; }
#pragma unroll
                        for (int j = 0; j < NNODES; ++j) a += lv[j] * xq[j];
#pragma unroll
                        for (int i = 0; i < NX; ++i) a += bv[i] * xb[i];
#pragma unroll
                        for (int j = 0; j < NNODES; ++j) a += (dv[j] - lv[j]) * xq[j];
#pragma unroll
                        for (int i = NX; i < NX + NU; ++i) a += bv[i] * xb[i];
                        if constexpr (NPAR > 0) a += arp * xs[P0];   // the last column
                        axz = isC ? a : 0.0;
                    }
                    (void)Dlo; (void)DtTlo;
                    sched_fence();
                } else {
                    // dense chains from the workspace
#pragma unroll
                    for (int e = 0; e < SL; ++e) {
                        double a = 0.0;
                        for (int k = 0; k < MM; ++k) a += Acol(k, e, zr) * bcast_uniform(ya, k);
                        aty[e] = isP[e] ? a : 0.0;
                    }
                    double a = 0.0;
                    for (int j = 0; j < NN; ++j) {
                        unsigned b = (unsigned)(j * N + NN) + (unsigned)rc + (unsigned)zr; asm("" : "+v"(b));
                        a += H[b] * ((j < 64) ? bcast_uniform(xv[0], j) : bcast_uniform(xv[1], j - 64));
                    }
                    axz = isC ? a : 0.0;
                }
                // H x: the rows of the first slot — RCS loads in flight per batch
                if (!ident) {
                    double hx = 0.0;
#pragma unroll
                    for (int j0 = 0; j0 < NN; j0 += RCS) {
                        double mm[RCS];
#pragma unroll
                        for (int j = 0; j < RCS; ++j) {
                            const unsigned l = lane_near(zr);
                            unsigned b = l + (unsigned)(((j0 + j < NN) ? j0 + j : 0) * N) + (unsigned)zr; asm("" : "+v"(b));
                            mm[j] = H[b];
                        }
#pragma unroll
                        for (int j = 0; j < RCS; ++j) if (j0 + j < NN) hx += mm[j] * xbc(xv, j0 + j);
                        sched_fence();
                    }
                    acc[0] = hx;
                }
                if (ident) {
                } else if constexpr (SMALL) {
                } else if constexpr (!FEW1) {   // many primal rows in the second slot: lane l < NP1 loads row 64 + l (the other lanes re-read row 64)
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
                    acc[1] = (sl < NP1) ? hx1 : 0.0;
                } else {
                    // rows 64 .. NN-1 of H: lane j loads H(64 + t, j) (and lane j < NP1 also H(64 + t, 64 + j)) and forms the product with its own x_j;
                    // lane t then adds the NN products of row 64 + t in ascending j — instead of NN loads per lane for NP1 live lanes
                    double* pb = tr + CD::PB_OFF;
                    double h0[NP1 > 0 ? NP1 : 1], h1[NP1 > 0 ? NP1 : 1];
#pragma unroll
                    for (int t = 0; t < NP1; ++t) {
                        const unsigned l = lane_near(zr);
                        unsigned b0 = (64u + (unsigned)t) + l * (unsigned)N + (unsigned)zr; asm("" : "+v"(b0));
                        h0[t] = H[b0];
                        unsigned b1 = (64u + (unsigned)t) + (64u + (l < (unsigned)NP1 ? l : 0u)) * (unsigned)N + (unsigned)zr; asm("" : "+v"(b1));
                        h1[t] = H[b1];
                    }
#pragma unroll
                    for (int t = 0; t < NP1; ++t) { pb[t * NN + sl] = h0[t] * xv[0]; if (sl < NP1) pb[t * NN + 64 + sl] = h1[t] * xv[1]; }
                    lds_order();
                    const int tt = sl < NP1 ? sl : 0;
                    double sacc = 0.0;
#pragma unroll
                    for (int j0 = 0; j0 < NN; j0 += 36) {
                        double pv[36];
#pragma unroll
                        for (int j = 0; j < 36; ++j) pv[j] = pb[tt * NN + ((j0 + j < NN) ? j0 + j : 0)];
#pragma unroll
                        for (int j = 0; j < 36; ++j) if (j0 + j < NN) sacc += pv[j];
                    }
                    acc[1] = (sl < NP1) ? sacc : 0.0;
                    lds_order();
                }
                double a1 = isC ? fmax(fabs(axz), fabs(zv)) : 0.0, a2 = 0.0, rp = isC ? fabs(axz - zv) : 0.0, rq = 0.0, rd = 0.0;
#pragma unroll
                for (int e = 0; e < SL; ++e) {
                    a1 = fmax(a1, isP[e] ? fabs(xv[e]) : 0.0);
                    a2 = fmax(a2, isP[e] ? fmax(fmax(fabs(acc[e]), fabs(aty[e])), fmax(fabs(hv[e]), fabs(yb[e]))) : 0.0);
                    rq = fmax(rq, isP[e] ? fabs(xv[e] - qv[e]) : 0.0);
                    rd = fmax(rd, isP[e] ? fabs(((acc[e] + hv[e]) + aty[e]) + yb[e]) : 0.0);
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
                    for (int e = 0; e < SL; ++e) {
                        const double prev = rhob[e];
                        rhob[e] = rho_of(typ[e], rho);
                        rhobinv[e] = 1.0 / rhob[e];
                        kd[e] = isP[e] ? (kd[e] + (rhob[e] - prev)) : 0.0;   // update_kkt_rho, box_admm.hpp:448-452
                    }
                    rhoc = rho_of(ctyp, rho); rhocinv = 1.0 / rhoc;
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
    for (int e = 0; e < SL; ++e) if (isP[e]) { out_x[lp[e]] = xv[e]; out_y[MM + lp[e]] = yb[e]; }
    if (isC) out_y[rc] = ya;
    const bool bad = __builtin_amdgcn_ballot_w64((((xv[0] - xv[0]) + (yb[0] - yb[0])) + ((xv[1] - xv[1]) + (yb[1] - yb[1])) + (ya - ya)) != 0.0) != 0;   // non-finite x or y
    info.status = status; info.iter = iter; info.rho_updates = rho_updates; info.flags = bad ? PMPC_FLAG_NONFINITE : 0;
    info.rho_estimate = rho_estimate; info.res_prim = res_prim; info.res_dual = res_dual;
}

}  // namespace pmpc
