// polympc_amd — SQP / linearisation kernels and their launch templates. Included by pmpc_api.hip for the built-in OCPs
// and by include/polympc/register_ocp.hpp for user-defined OCPs (compiled by hipcc in the user's translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../include/polympc_amd.h"
#include "pmpc_ocp.hpp"
#include "pmpc_qp.hpp"
#include "pmpc_qp_reg.hpp"
#include "pmpc_qp_reg2.hpp"
#include "pmpc_qp_cond.hpp"
#include "pmpc_qp_big.hpp"
#include "pmpc_sqp.hpp"

// context services exported by libpolympc_amd.so (collocation constants cache, HBM workspace, stream, limits)
extern "C" pmpc_status pmpc_internal_services(pmpc_context* ctx, int P, int S, double t0, double tf, size_t ws_bytes, const void** cheb,
                                               double** ws, void** stream, size_t* lds_limit, unsigned long long** phase_cycles, int* force_lds);
extern "C" int pmpc_internal_simd_count(pmpc_context* ctx);   // SIMDs of the device (compute units x 4)
extern "C" int pmpc_internal_sqp_slice(pmpc_context* ctx);   // SQP iterations per kernel launch (0 = whole solve in one launch)
extern "C" int pmpc_internal_sqp_rr(pmpc_context* ctx);      // 1 (PMPC_SQP_RR=1, developer switch): batches beyond the resident wavefronts run one SQP iteration per work item (sqp_kernel_rr)
extern "C" void pmpc_internal_set_route(pmpc_context* ctx, int route);   // records the kernel family of the launch (pmpc_sqp_last_route)
extern "C" int pmpc_internal_last_route(pmpc_context* ctx);
// developer switches of the launcher, read from the environment ONCE per context (pmpc_create): PMPC_NO_REDO_LAUNCH (timing the redo launches), PMPC_NO_CONDREG,
// PMPC_NO_SCHUR (keep the dense kernels), PMPC_SCHUR_SMALL (block-structured kernel on at most 64 KKT rows), PMPC_BIG_WG4 = 0 / 1 (team kernel never / whenever eligible)
enum { PMPC_SW_NO_REDO_LAUNCH = 0, PMPC_SW_NO_CONDREG = 1, PMPC_SW_NO_SCHUR = 2, PMPC_SW_SCHUR_SMALL = 3, PMPC_SW_BIG_WG4_ON = 4, PMPC_SW_BIG_WG4_OFF = 5, PMPC_SW_NO_CONDREG_RUIZ = 6 };
extern "C" int pmpc_internal_switch(pmpc_context* ctx, int which);

namespace pmpc {
using ::pmpc_status;

constexpr int FILTER_LDS_DOUBLES = 24;   // PMPC_FILTER_STATE_DOUBLES rounded up

// LDS staging (doubles) of the register-resident QP that serves a compile-time size: one row per lane up to 64 KKT rows, two rows per lane up to 112
template <int NKKT> constexpr int reg_qp_staging() {
    if constexpr (NKKT <= 0) return 0;
    else if constexpr (NKKT <= WAVE) return RegKkt<NKKT>::TRI;
    else return RegKkt2<NKKT>::TRI;
}

// the same for the CONDENSED register QP (pmpc_qp_cond.hpp) of NN variables, MM rows on NNODES nodes: its tile set's staging (the one-row-per-lane set up to 64
// variables; beyond, RegKkt2<NN, PMPC_COND_NV> — NOT the (NN + MM)-row full inverse's, whose LDS-resident operand tiles the condensed kernel does not have) or
// the exchange vectors + D~ tables that alias it, whichever is larger. (Until round 5 the condensed kernels were launched with the full inverse's staging: the
// 16-node robot grid took 53.7 KB per instance — THREE workgroups per CU, a SIMD idle — where 33.6 KB do.)
template <int NN, int MM, int NNODES> constexpr int cond_qp_staging() {
    if constexpr (NN <= WAVE) return RegKkt<NN>::TRI;
    else {
        constexpr int a = RegKkt2<NN, PMPC_COND_NV>::TRI, b = CondDims<NN, MM>::TAB_OFF + CondDims<NN, MM>::template tab_doubles<NNODES>();
        return a > b ? a : b;
    }
}

// the hook builds (POL): D~ tables per state index from the workspace (pmpc_qp_cond.hpp WS) — 4 NX tables instead of 4, which no longer fit the sweep's staging
template <int NN, int MM, int NNODES, int NX> constexpr int cond_qp_staging_ws() {
    constexpr int a = cond_qp_staging<NN, MM, NNODES>(), b = CondDims<NN, MM>::TAB_OFF + CondDims<NN, MM>::template tab_doubles_ws<NNODES, NX>();
    return a > b ? a : b;
}

// LDS doubles of the block-sparse copy of J the register-resident kernels keep (pmpc_jview.hpp): per node NX x NDER + NG x NDER
template <class Model> __host__ __device__ inline size_t jview_doubles(int nnodes) {
    return (size_t)nnodes * (Model::NX + Model::NG) * OcpDims<Model>::JBS;
}

// large-instance mode: per-instance HBM scratch (doubles) behind the factor workspace — SQP vectors, per-node AD staging, QP vectors
template <class Model> __host__ __device__ inline size_t big_scratch_doubles(int P, int S) {
    OcpDims<Model> dm(P, S);
    return ((SqpLds::doubles(dm.n, dm.m, dm.mi) + OcpLds<Model>::doubles(P, S) + QpLds::doubles_rest(dm.n, dm.m) + jview_doubles<Model>(dm.NN) + 8) + 1) & ~(size_t)1;   // (even: every instance's factor workspace starts on a 16-byte boundary — its panels are read 16 bytes per lane)
}

// KHBM: large-instance mode of the LDS-resident kernels — the KKT factor lives in an HBM workspace (Kws). A compile-time flag so
// that in the normal mode every QP pointer provably addresses LDS (ds_read / ds_write instead of flat accesses, which cost the
// LDS path most of its time when the location of K was a run-time choice)
template <class Model, int NN = 0, int MM = 0, bool PROF = false, int HU = 0, bool KHBM = false, bool W2 = false, bool POL = false, bool CND = false, bool WG4 = false>   // WG4: the HBM-factor kernel on a workgroup of FOUR wavefronts per instance (BigTeam, pmpc_qp_big.hpp: small batches / lone instances); CND: condensed register QP (pmpc_qp_cond.hpp); W2: the HBM-factor kernel compiled for two wavefronts per SIMD (256 registers); POL: register-resident kernel with the Ruiz / filter-line-search hooks compiled in
#ifndef PMPC_SQP_WAVES
#define PMPC_SQP_WAVES 2
#endif
#ifndef PMPC_BIG_WAVES
#define PMPC_BIG_WAVES 1
#endif
#ifndef PMPC_COND1_WAVES
#define PMPC_COND1_WAVES 1   /* condensed register kernel on at most 64 variables: wavefronts per SIMD — measured on the 11-node robot grid: two wavefronts (256 registers: 240 spilled values, eight scratch accesses in the ADMM loop) 2.82 ms per 4096, one wavefront 2.67 (the full inverse: 4.04) */
#endif
__global__ __launch_bounds__(WG4 ? 256 : 64, ((NN > 0 && NN + MM <= 64) ? PMPC_SQP_WAVES : ((NN > 0 && CND && NN <= 64) ? PMPC_COND1_WAVES : ((KHBM && W2) ? 2 : (KHBM ? PMPC_BIG_WAVES : 1))))) void sqp_kernel(Model model, const ChebData* __restrict__ cd, int B,
                                                 const double* __restrict__ x_guess, const double* __restrict__ lam_guess,
                                                 const double* __restrict__ d, const double* __restrict__ lbx,
                                                 const double* __restrict__ ubx, const double* __restrict__ lbg,
                                                 const double* __restrict__ ubg, pmpc_sqp_settings ss, pmpc_qp_settings qs,
                                                 double* __restrict__ Hws, double* __restrict__ Aws, double* __restrict__ x,
                                                 double* __restrict__ lam, pmpc_sqp_info* __restrict__ info, unsigned long long* __restrict__ phase_cycles,
                                                 double* __restrict__ Kws, int it_begin, int it_end, double* __restrict__ slice_state, unsigned lds_dyn_doubles) {
    extern __shared__ double smem[];
    const size_t lds_doubles_total = __builtin_amdgcn_groupstaticsize() / sizeof(double) + lds_dyn_doubles;
    const int b = blockIdx.x;
    if (b >= B) return;
    // redo launch (the full-KKT-form kernel behind a condensed one): only the instances whose condensed solve gave up at its conditioning gate
    int flags0 = 0;   // (redo launch: PMPC_FLAG_ILLCOND is part of the instance's flags from the start — information for the caller: this instance took the full KKT form)
    if (it_begin == PMPC_REDO_MODE) { if (__builtin_amdgcn_readfirstlane(info[b].status) != PMPC_SQP_REDO) return; it_begin = 0; flags0 = PMPC_FLAG_ILLCOND; }
    // iteration-sliced execution: instances that finished in an earlier slice give their slot back immediately
    if (it_begin != 0 && __builtin_amdgcn_readfirstlane(info[b].status) != PMPC_SQP_IN_PROGRESS) return;
    // it_begin < 0: resume mode (the launch behind the round-robin kernel, sqp_kernel_rr below): every instance continues from the iteration its own record holds
    if (it_begin < 0) it_begin = __builtin_amdgcn_readfirstlane(info[b].iter);
    const int P = cd->P, S = cd->S;
    Ocp<Model> ocp(model, P, S, cd->t_scale);
    const int n = ocp.dm.n, m = ocp.dm.m, mi = ocp.dm.mi;
    QpLds qw; SqpLds v;
    // KHBM, large-instance mode: the KKT factor lives in an HBM workspace, and so does everything else except the vectors the substitutions
    // hammer (the QP solution x / y and the right-hand side): the SQP vectors (20 of length n or n+m), the per-node AD staging and the remaining
    // QP vectors go to a per-instance scratch region behind the factor. With them in LDS one instance took 157 KB — ONE wavefront per CU, three
    // SIMDs of four idle (config C: 256 instances in flight, four rounds); now a CU holds one instance per SIMD.
    double* p; double* stage0; const double* stage_end; double* big_mail = nullptr;
    if constexpr (NN == 0 && KHBM) {
        p = qw.carve_xy(smem, n, m);
        double* rhsL = p; p += n + m;
        qw.big_lds = p; p += BigKkt::LDS_DOUBLES;   // diagonal tile + broadcast slots of the blocked factorisation
        big_mail = p; p += BIG_MAIL_DOUBLES;        // mailbox of the four-wavefront team (WG4)
        ocp.Dlds = p; p += (size_t)(P + 1) * (P + 2);
        if constexpr (Model::NG == 0 && Model::NP == 0) { if (JViewRT<Model>::tab_worth_it(ocp.dm.NN)) { p += (p - smem) & 1; ocp.jtab = p; p += JViewRT<Model>::tab_doubles(ocp.dm.NN); } }   // D~ tables of the condensed solve's sparse products
        double* Wb = Kws + (size_t)b * (BigKkt::doubles(n + m) + big_scratch_doubles<Model>(P, S));
        double* hq = Wb + BigKkt::doubles(n + m);
        hq = v.carve(hq, n, m, mi);
        stage0 = hq;
        hq = ocp.s.carve(hq, P, S);
        stage_end = hq;
        qw.carve_rest_split(hq, rhsL, n, m, Wb);
        hq += QpLds::doubles_rest(n, m);
        ocp.jblk = hq; ocp.gblk = ocp.jblk + (size_t)ocp.dm.NN * Model::NX * OcpDims<Model>::JBS; ocp.keep_blk = true;   // block-sparse copy of J for the condensed linear algebra (pmpc_qp_big.hpp)
    } else {
        p = (NN > 0) ? qw.carve_xy(smem, n, m) : qw.carve(smem, n, ss.qp_solver == 1 ? m + n : m);   // ADMM: stacked constraint rows
        p = v.carve(p, n, m, mi);
        stage0 = p;
        p = ocp.s.carve(p, P, S);
        constexpr int QP_STAGING0 = []() constexpr {   // (condensed register QP: its own tile set's staging)
            if constexpr (CND) return cond_qp_staging<NN, MM, NN / (Model::NX + Model::NU)>();
            else return reg_qp_staging<NN + MM>(); }();
        int QP_STAGING = QP_STAGING0;
        if constexpr (CND && POL) {   // its hook builds with the Ruiz preconditioner on: room for the tables per state index (the launcher sized the allocation by the same rule)
            if (ss.preconditioner == 1) QP_STAGING = cond_qp_staging_ws<NN, MM, NN / (Model::NX + Model::NU), Model::NX>();
        }
        if (NN > 0 && (size_t)(p - stage0) < (size_t)QP_STAGING + 2 + ocp.s.const_doubles(P, S)) p = stage0 + QP_STAGING + 2 + ocp.s.const_doubles(P, S);
        stage_end = p;   // end of the per-node staging block: what follows (static parameters, filter) stays live during the line search
    }
    if constexpr (NN > 0 && (NN + MM > WAVE || CND)) {   // block-sparse copy of J (pmpc_jview.hpp) for the two-rows-per-lane and the condensed kernels: lives through the QP, behind the staging its LDS buffers alias
        ocp.jblk = p; p += jview_doubles<Model>(ocp.dm.NN); ocp.gblk = ocp.jblk + (size_t)ocp.dm.NN * Model::NX * OcpDims<Model>::JBS; ocp.keep_blk = true;
    }
    double* dL = p; p += (Model::ND > 0 ? Model::ND : 1);
    const int ln = lane_id();
    ocp.d = dL;
    double* filt = nullptr;   // LSFilter of this instance (line_search = 1; the register-resident specialisations do not carry it)
    if constexpr (NN == 0 || POL) { filt = p; p += FILTER_LDS_DOUBLES; }
    if constexpr (WG4) {
        // wavefronts 1 .. 3 of a four-wavefront team (BigTeam): every pointer above is set up on them as on the first wavefront — the carve is pure
        // arithmetic —, no data has moved yet; from here on they run the routines the first wavefront posts (big_helper_loop) and nothing else
        if (threadIdx.x >= WAVE) { big_helper_loop<Model>((BigMail<JViewRT<Model>>*)big_mail, (int)(threadIdx.x >> 6), ocp); return; }
    }
    for (int i = ln; i < Model::ND; i += WAVE) dL[i] = d[(size_t)b * Model::ND + i];
    if constexpr (NN == 0 || POL) {
        const bool carried = ss.line_search == 1 && ss.filter_state != nullptr;
        if (ln < PMPC_FILTER_STATE_DOUBLES) filt[ln] = carried ? ss.filter_state[(size_t)b * PMPC_FILTER_STATE_DOUBLES + ln] : 0.0;
    }
    double* eigw = nullptr;   // eigenvalue-mirroring regulariser (regularisation = 1): A and V of the Jacobi iteration, 2 n^2 doubles; allocated on request only
    if constexpr (NN == 0 || POL) { if (ss.regularisation == 1) { eigw = p; p += 2 * (size_t)n * n; } }   // (round 6: the hook builds of the register kernels carry the policy too)
    ocp.stage_constants(cd);
    if constexpr (NN == 0 && KHBM) {
        for (int i = ln; i < (P + 1) * (P + 2); i += WAVE) ocp.Dlds[i] = ocp.s.D[i];
        wsync();
        if constexpr (Model::NG == 0 && Model::NP == 0) { if (JViewRT<Model>::tab_worth_it(ocp.dm.NN)) JViewRT<Model>::build_tables(ocp.Dlds, P, ocp.dm.NN, ocp.jtab); }
    }
    double* sst = slice_state ? slice_state + (size_t)b * 2 * n : nullptr;   // [previous Lagrangian gradient | previous step]
    // Conditioning rule of the register-resident kernels that eliminate the constraint block first (PMPC_FLAG_ILLCOND, include/polympc_amd.h), decided per instance
    // from its BOUNDS and BEFORE any work (round 6; ADVICE r5: the one-row-per-lane kernels used to test behind the solve and throw a finished solve away): these
    // kernels invert S = P + A' diag(rho) A (constraint-first sweep / condensed form), whose condition number rho_eq |A|^2 / lambda_min(P on null A) stays ~1e5
    // whatever rho is as long as the directions the collocation Jacobian leaves free — the controls and parameters — are BOUNDED (rho_box scales with rho), and
    // grows with rho when one of them is not (rho_box = RHO_MIN). Such an instance is handed to the redo launch (full KKT form) untouched. The test rides on the
    // loop that loads the bounds — two compares on values that are in registers anyway; round 5's separate loop over the LDS copies (130 instructions) cost the
    // bench kernel 2 % in front of the solve. A numeric gate at every factorisation — what the large-instance kernel and the QP entry point use — cost these
    // kernels 4 .. 10 % through register pressure alone (EXPERIMENTS.md round 5).
    constexpr bool BOUNDS_RULE = NN > 0 && (CND || NN + MM <= WAVE);
    bool loose = false;
    for (int i = ln; i < n; i += WAVE) {
        v.x[i] = (it_begin > 0) ? x[(size_t)b * n + i] : (x_guess ? x_guess[(size_t)b * n + i] : 0.0);
        const double lbi = lbx[(size_t)b * n + i], ubi = ubx[(size_t)b * n + i];
        v.lbx[i] = lbi; v.ubx[i] = ubi;
        if constexpr (BOUNDS_RULE) loose |= (i >= ocp.dm.VARX) && (lbi < -LOOSE_BOUNDS_THRESH) && (ubi > LOOSE_BOUNDS_THRESH);   // classify_bounds(...) == 2 on a control / parameter
        if (it_begin > 0) { v.lg[i] = sst[i]; v.step[i] = sst[n + i]; }
    }
    for (int i = ln; i < m + n; i += WAVE)
        v.lam[i] = (it_begin > 0) ? lam[(size_t)b * (m + n) + i] : (lam_guess ? lam_guess[(size_t)b * (m + n) + i] : 0.0);
    for (int i = ln; i < mi; i += WAVE) {
        v.lbg[i] = lbg ? lbg[(size_t)b * mi + i] : -INFINITY;
        v.ubg[i] = ubg ? ubg[(size_t)b * mi + i] : INFINITY;
    }
    wsync();
    if constexpr (BOUNDS_RULE) {
        if (it_begin == 0 && flags0 == 0 && __builtin_amdgcn_ballot_w64(loose) != 0) {
            // (x and lam leave as the guesses: the outputs of an instance that gave up are defined even when a developer's PMPC_NO_REDO_LAUNCH skips the launch that re-solves it)
            for (int i = ln; i < n; i += WAVE) x[(size_t)b * n + i] = v.x[i];
            for (int i = ln; i < m + n; i += WAVE) lam[(size_t)b * (m + n) + i] = v.lam[i];
            if (ln == 0) { pmpc_sqp_info r; r.iter = 0; r.qp_solver_iter = 0; r.status = PMPC_SQP_REDO; r.flags = PMPC_FLAG_ILLCOND; r.primal_norm = 0.0; r.dual_norm = 0.0; r.max_violation = 0.0; r.cost = 0.0; info[b] = r; }
            return;
        }
    }
    // stacked workspace K0 = [H ; J] ((n+m) x n, column-major, leading dimension n+m): lane i reads row i of K0 with ONE stride
    double* K0 = Hws + (size_t)b * (size_t)(n + m) * n;
    (void)Aws;
    SqpDevice<Model, NN, MM, PROF, HU, KHBM, POL, CND ? -1 : ((KHBM && WG4) ? -3 : ((KHBM && W2) ? -2 : 0))> sqp(ocp, v, qw, K0, K0 + n, ss, qs);   // (W2 && WG4: the team kernel built for 256 registers — two workgroups per CU)
    sqp.big_mail = big_mail;
    sqp.filt = filt;
    sqp.eig = eigw;
    sqp.trace = ss.iteration_trace ? ss.iteration_trace + (size_t)b * (size_t)ss.iteration_trace_capacity * PMPC_TRACE_DOUBLES : nullptr;
    sqp.tr = ocp.s.fval;   // first per-node staging array: everything from here on is dead while the QP runs
    if constexpr (NN + MM > 112) sqp.tr += (ocp.s.fval - smem) & 1;   // (on a 16-byte boundary there: the 8 x 8 register path reads its LDS tiles in 16-byte units. Only there — a second, separately live pointer cost the 7-node kernel a spill inside its ADMM loop)
    {   // side-by-side line search: G candidates x (m constraint values + NN Lagrange values) + 3 scalars each, in the same region
        const int G = WAVE / ocp.dm.NN;
        const size_t need = (size_t)G * (m + ocp.dm.NN + 3);
        const size_t have = (size_t)(stage_end - ocp.s.fval);   // the staging block only — never the parameters / filter carved behind it
        sqp.lsbuf = ocp.s.fval;   // a flag, not a null pointer, says whether it fits (compiler hazard 7: null tests of LDS pointers)
        sqp.ls_side_by_side = (G >= 2 && need <= have);
    }
    pmpc_sqp_info si;
    sqp.qp_flags = flags0;
    if (it_begin > 0) { const pmpc_sqp_info prev = info[b]; sqp.qp_iter_total = prev.qp_solver_iter; sqp.qp_flags = prev.flags; sqp.cost_log = prev.cost; }
    sqp.solve(si, it_begin, it_end);
    if (si.status == PMPC_SQP_IN_PROGRESS && sst) for (int i = ln; i < n; i += WAVE) { sst[i] = v.lg[i]; sst[n + i] = v.step[i]; }
    for (int i = ln; i < n; i += WAVE) x[(size_t)b * n + i] = v.x[i];
    for (int i = ln; i < m + n; i += WAVE) lam[(size_t)b * (m + n) + i] = v.lam[i];
    if (ln == 0) info[b] = si;
    if constexpr (NN == 0 || POL) {   // (an instance that gave up leaves the carried filter as it found it: the redo launch starts from the same filter)
        if (ss.line_search == 1 && ss.filter_state != nullptr && si.status != PMPC_SQP_REDO && ln < PMPC_FILTER_STATE_DOUBLES) ss.filter_state[(size_t)b * PMPC_FILTER_STATE_DOUBLES + ln] = filt[ln];
    }
    if constexpr (PROF) { if (phase_cycles && ln == 0) for (int i = 0; i < 24; ++i) atomicAdd(&phase_cycles[i], (unsigned long long)sqp.cyc[i]); }
    if constexpr (WG4) {   // release the helpers
        if (ln == 0) ((BigMail<JViewRT<Model>>*)big_mail)->op = BIG_OP_EXIT;
        __syncthreads();
    }
}
// ---------------------------------------------------------------------------------------------------------------------------------------------
// Block-structured specialisation (pmpc_qp_schur.hpp): the Hessian is block diagonal per node — the block BFGS every control test of the reference
// selects (continuous_ocp.hpp:2304-2431) or exact Hessians, NG = 0, at most one parameter (arrow shape: a border row / column beside the blocks) — and lives with J's per-node blocks in LDS: no HBM workspace, the QP through
// the m x m Schur complement. One instantiation per (model, P, S): the segment structure is a compile-time constant of the sparse products.
template <class Model, int PP, int SS> constexpr int schur_lds_doubles_ct() {
    using SD = SchurDims<Model, PP, SS>;
    return SD::LDS_DOUBLES + SD::NNODES * SD::DD /*hblk*/ + (SD::NPAR ? 2 * SD::N : 0) /*hbrd*/ + SD::NNODES * SD::NX * SD::JBS /*jblk*/;
}
template <class Model, int PP, int SS> inline size_t sqp_schur_lds_bytes(bool pol = false) {   // (exactly what sqp_schur_kernel carves + 2: at 16 robot nodes 40 952 bytes — four instances per CU; 984 more were three)
    using SD = SchurDims<Model, PP, SS>;
    OcpDims<Model> dm(PP, SS);
    size_t stage = OcpLds<Model>::doubles(PP, SS);
    const size_t need = (size_t)RegKkt<SD::M>::TRI + OcpLds<Model>::const_doubles(PP, SS) + 8;
    if (stage < need) stage = need;
    return ((size_t)(2 * dm.n + dm.m) + SqpLds::doubles(dm.n, dm.m, dm.mi) + stage + schur_lds_doubles_ct<Model, PP, SS>() + (Model::ND > 0 ? Model::ND : 1) + (pol ? FILTER_LDS_DOUBLES : 0) + 2) * sizeof(double);
}
template <class Model, int PP, int SS, bool PROF = false, bool POL = false>   // POL: with the filter line search (line_search = 1, LSFilter carried in LDS) — the hook of the reference's tests this kernel family carries
__global__ __launch_bounds__(64, (SchurDims<Model, PP, SS>::N + SchurDims<Model, PP, SS>::M <= 64) ? PMPC_SQP_WAVES : 1)
void sqp_schur_kernel(Model model, const ChebData* __restrict__ cd, int B, const double* __restrict__ x_guess, const double* __restrict__ lam_guess,
                      const double* __restrict__ d, const double* __restrict__ lbx, const double* __restrict__ ubx, pmpc_sqp_settings ss,
                      pmpc_qp_settings qs, double* __restrict__ x, double* __restrict__ lam, pmpc_sqp_info* __restrict__ info,
                      unsigned long long* __restrict__ phase_cycles) {
    using SD = SchurDims<Model, PP, SS>;
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    if (b >= B) return;
    Ocp<Model> ocp(model, PP, SS, cd->t_scale);
    constexpr int n = SD::N, m = SD::M;
    QpLds qw; SqpLds v;
    double* p = qw.carve_xy(smem, n, m);
    p = v.carve(p, n, m, 0);
    double* stage0 = p;
    p = ocp.s.carve(p, PP, SS);
    if ((size_t)(p - stage0) < (size_t)RegKkt<m>::TRI + 2 + ocp.s.const_doubles(PP, SS)) p = stage0 + RegKkt<m>::TRI + 2 + ocp.s.const_doubles(PP, SS);
    const double* stage_end = p;
    ocp.jblk = p; p += SD::NNODES * SD::NX * SD::JBS; ocp.gblk = ocp.jblk; ocp.keep_blk = true;
    double* hblk = p; p += SD::NNODES * SD::DD;
    double* hbrd = p; p += (SD::NPAR ? 2 * SD::N : 0);
    double* qblk = p; p += SD::NNODES * SD::DD;
    double* xsc = p; p += n + 1;
    double* dsc = p; p += m + 1;
    p += ((p - smem) & 1);   // the tables are read 16 bytes at a time
    double* dtab = p; p += SD::TAB;
    double* dL = p; p += (Model::ND > 0 ? Model::ND : 1);
    double* filt = nullptr;   // LSFilter of this instance (POL)
    if constexpr (POL) { filt = p; p += FILTER_LDS_DOUBLES; }
    const int ln = lane_id();
    for (int i = ln; i < Model::ND; i += WAVE) dL[i] = d[(size_t)b * Model::ND + i];
    if constexpr (POL) {
        const bool carried = ss.line_search == 1 && ss.filter_state != nullptr;
        if (ln < PMPC_FILTER_STATE_DOUBLES) filt[ln] = carried ? ss.filter_state[(size_t)b * PMPC_FILTER_STATE_DOUBLES + ln] : 0.0;
    }
    ocp.d = dL;
    ocp.stage_constants(cd);
    for (int i = ln; i < n; i += WAVE) {
        v.x[i] = x_guess ? x_guess[(size_t)b * n + i] : 0.0;
        v.lbx[i] = lbx[(size_t)b * n + i]; v.ubx[i] = ubx[(size_t)b * n + i];
    }
    for (int i = ln; i < m + n; i += WAVE) v.lam[i] = lam_guess ? lam_guess[(size_t)b * (m + n) + i] : 0.0;
    wsync();
    SqpDevice<Model, n, m, PROF, 1, false, POL, PP * 256 + SS> sqp(ocp, v, qw, nullptr, nullptr, ss, qs);
    sqp.filt = filt;
    sqp.hblk = hblk; sqp.hbrd = hbrd; sqp.qblk = qblk; sqp.xsc = xsc; sqp.dsc = dsc; sqp.dtab = dtab;
    schur_build_tables<Model, PP, SS>(ocp.s.D, dtab);
    sqp.trace = ss.iteration_trace ? ss.iteration_trace + (size_t)b * (size_t)ss.iteration_trace_capacity * PMPC_TRACE_DOUBLES : nullptr;
    sqp.tr = ocp.s.fval;
    {
        const int G = WAVE / SD::NNODES;
        const size_t need = (size_t)G * (m + SD::NNODES + 3);
        const size_t have = (size_t)(stage_end - ocp.s.fval);
        sqp.lsbuf = ocp.s.fval;
        sqp.ls_side_by_side = (G >= 2 && need <= have);
    }
    pmpc_sqp_info si;
    sqp.solve(si, 0, ss.max_iter);
    for (int i = ln; i < n; i += WAVE) x[(size_t)b * n + i] = v.x[i];
    for (int i = ln; i < m + n; i += WAVE) lam[(size_t)b * (m + n) + i] = v.lam[i];
    if (ln == 0) info[b] = si;
    if constexpr (POL) {   // (an instance that gave up leaves the carried filter as it found it: the redo launch starts from the same filter)
        if (ss.line_search == 1 && ss.filter_state != nullptr && si.status != PMPC_SQP_REDO && ln < PMPC_FILTER_STATE_DOUBLES) ss.filter_state[(size_t)b * PMPC_FILTER_STATE_DOUBLES + ln] = filt[ln];
    }
    if constexpr (PROF) { if (phase_cycles && ln == 0) for (int i = 0; i < 24; ++i) atomicAdd(&phase_cycles[i], (unsigned long long)sqp.cyc[i]); }
}
// the request can take the block-structured kernel: block-diagonal Hessian throughout, default policies otherwise (pmpc_sqp_settings::kkt_form = 1
// asks for the reference's full KKT form and keeps the dense kernels)
inline bool schur_request_ok(const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, int slice_iters) {
    return (ss->hessian_update == 1 || ss->exact_hessian_every_iter) && (ss->regularisation == 0 || ss->regularisation == 2) && ss->preconditioner == 0 &&
           ss->qp_solver == 0 && (ss->kkt_form == 0 || ss->kkt_form == 2) && qs->linear_solver == 0 && slice_iters == 0;   // (line_search = 1: the grids compiled with the hook, try_launch_schur)
}
template <class Model, int PP, int SS>
inline bool try_launch_schur(pmpc_context* ctx, const Model& mdl, const ChebData* cd, int P, int S, int B, const double* x_guess, const double* lam_guess,
                             const double* d, const double* lbx, const double* ubx, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x,
                             double* lam, pmpc_sqp_info* info, hipStream_t stream, size_t lds_limit, unsigned long long* phase, pmpc_status* st) {
    if (P != PP || S != SS) return false;
    // Systems of at most 64 KKT rows stay on the one-row-per-lane dense kernel: its ADMM iteration is ONE register mat-vec (600 cycles), the
    // block-structured one a chain of eight LDS exchanges and two mat-vecs (3.7 k cycles on config A) — measured 1.69 against 1.16 ms per 4096
    // config-A instances although the factorisation is three times cheaper. PMPC_SCHUR_SMALL=1: developer switch.
    if (SchurDims<Model, PP, SS>::N + SchurDims<Model, PP, SS>::M <= WAVE && !pmpc_internal_switch(ctx, PMPC_SW_SCHUR_SMALL)) return false;
    const size_t lds = sqp_schur_lds_bytes<Model, PP, SS>(ss->line_search == 1);
    if (lds > lds_limit) return false;
    auto kern = sqp_schur_kernel<Model, PP, SS, false>;
    if constexpr (SchurDims<Model, PP, SS>::NPAR == 0) { if (phase) kern = sqp_schur_kernel<Model, PP, SS, true>; }   // (no phase-timer build of the bordered form: a developer switch already)
    else phase = nullptr;
    // the filter line search (round 5): compiled for the grids of the reference's own tests with more than 64 KKT rows (11 and 16 nodes), no phase timers
    if (ss->line_search == 1) {
        if constexpr (SchurDims<Model, PP, SS>::NPAR == 0 && (SchurDims<Model, PP, SS>::NNODES == 11 || SchurDims<Model, PP, SS>::NNODES == 16)) { kern = sqp_schur_kernel<Model, PP, SS, false, true>; phase = nullptr; }
        else return false;
    }
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { *st = PMPC_ERR_HIP; return true; }
    pmpc_internal_set_route(ctx, PMPC_ROUTE_SCHUR);
    hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), lds, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, *ss, *qs, x, lam, info, phase);
    *st = (hipGetLastError() == hipSuccess) ? PMPC_OK : PMPC_ERR_HIP;
    return true;
}
// the grids with a block-structured specialisation, per model (defined in the model's own translation unit: pmpc_schur_*.hip)
template <class Model> struct SCHUR_GRIDS { static constexpr bool value = false; };
template <> struct SCHUR_GRIDS<RobotOCP> { static constexpr bool value = true; };
template <> struct SCHUR_GRIDS<CstrOCP> { static constexpr bool value = true; };
template <> struct SCHUR_GRIDS<ParkingOCP> { static constexpr bool value = true; };   // (NP = 1: the bordered form, round 5)
template <class Model>
bool try_launch_schur_grids(pmpc_context* ctx, const Model& mdl, const ChebData* cd, int P, int S, int B, const double* x_guess, const double* lam_guess,
                            const double* d, const double* lbx, const double* ubx, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x,
                            double* lam, pmpc_sqp_info* info, hipStream_t stream, size_t lds_limit, unsigned long long* phase, pmpc_status* st);

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Round-robin execution of a batch that exceeds the resident wavefronts (register-resident QP, one KKT row per lane) — a DEVELOPER SWITCH
// (PMPC_SQP_RR=1), measured and not made the default: see DESIGN.md §6. Instances need 4..max_iter SQP iterations, and with one workgroup per
// instance a launch of 4096 config-A instances takes 1.45 x its throughput time. Here the unit of work is ONE SQP ITERATION of one instance: a
// grid of resident wavefronts pops instances from a ready queue, runs the next iteration of the instance from its state in HBM (x, lam, the previous
// Lagrangian gradient and step, the BFGS matrix in the workspace: the state of the iteration-sliced launches), writes the state back and pushes
// the instance to the tail of the queue if it has to go on; an instance whose iteration took much longer than the wavefront's mean keeps the
// wavefront (it is on the critical path). The arithmetic of an iteration is the one-launch kernel's: bit-identical results
// (test_sqp_round_robin_execution_bit_identical). What the per-item time stamps of this kernel showed is why it does not pay: the launch is not
// waiting for instances that were dispatched late but for the serial work of the hardest ones (ten iterations whose QPs need several rho updates,
// i.e. re-factorisations: 2.1 M cycles = 0.98 ms under load for the worst of the 4096) — no order of execution shortens that.
//   Queues: instance b belongs to queue b % 8; a workgroup serves the queue of the XCD it finds itself on (s_getreg XCC_ID), so that the wavefront
//   that wrote an instance's state and the one that continues it share one L2. A queue is an array of nk x max_iter entries (an instance is
//   pushed at most max_iter times: no wrap-around), entry = (instance + 1) | (iterations done << 24), 0 = not written yet; head / tail / finished
//   counters on their own 128-byte lines. The producer drains its stores (s_waitcnt vmcnt(0): they have reached the XCD's L2) before the agent-scope
//   store of the entry; the consumer polls its entry with agent-scope loads and invalidates its CU's L1 (agent-scope acquire) before reading the state.
//   Placement changes speed only: the resume-mode launch of sqp_kernel behind this kernel (it_begin < 0) finishes whatever a queue without
//   workgroups, or a wavefront that gave up waiting (every spin is bounded), left in progress.
constexpr int RR_QUEUES = 8;
constexpr int RR_LINE = 32;                                  // ints per 128-byte line
constexpr int RR_HEADER_WORDS = RR_QUEUES * 3 * RR_LINE;     // per queue: head, tail, finished
constexpr int RR_PASS_SHIFT = 24;                            // entry = (b + 1) | (iterations done << 24)
constexpr int RR_MAX_BATCH = (1 << RR_PASS_SHIFT) - 2;
constexpr int RR_MAX_ITER = 127;
constexpr int RR_SPIN_LIMIT = 1 << 16;
#ifndef PMPC_RR_HARD_TENTHS
#define PMPC_RR_HARD_TENTHS 14
#endif
constexpr int RR_HARD_TENTHS = PMPC_RR_HARD_TENTHS;   // an iteration is 'hard' when it takes more than this many tenths of the wavefront's mean
inline size_t rr_queue_bytes(int B, int max_iter) { return ((size_t)RR_HEADER_WORDS + (size_t)B * (size_t)(max_iter > 0 ? max_iter : 1)) * sizeof(int); }
__host__ __device__ inline int rr_queue_len(int B, int q) { return (q < B) ? (B - q + RR_QUEUES - 1) / RR_QUEUES : 0; }   // instances q, q + 8, ...

template <class Model>   // (a template so that every translation unit may carry it)
__global__ __launch_bounds__(256) void sqp_rr_init_kernel(int* __restrict__ queue, pmpc_sqp_info* __restrict__ info, int B, int max_iter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { pmpc_sqp_info z; z.iter = 0; z.qp_solver_iter = 0; z.status = PMPC_SQP_IN_PROGRESS; z.flags = 0; z.primal_norm = 0.0; z.dual_norm = 0.0; z.max_violation = 0.0; z.cost = 0.0; info[i] = z; }
    if (i < RR_HEADER_WORDS) {   // head = 0, tail = number of instances of the queue (all of them are ready), finished = 0
        const int q = i / (3 * RR_LINE), w = i - q * 3 * RR_LINE;
        queue[i] = (w == RR_LINE) ? rr_queue_len(B, q) : 0;
    }
    if (i < B * max_iter) {
        int start = 0, val = 0;
        for (int q = 0; q < RR_QUEUES; ++q) {
            const int nk = rr_queue_len(B, q), cap = nk * max_iter;
            if (i >= start && i < start + cap) { const int k = i - start; val = (k < nk) ? (q + RR_QUEUES * k + 1) : 0; }
            start += cap;
        }
        queue[RR_HEADER_WORDS + i] = val;
    }
}

template <class Model, int NN, int MM, int HU>
__global__ __launch_bounds__(64, PMPC_SQP_WAVES) void sqp_kernel_rr(Model model, const ChebData* __restrict__ cd, int B,
                                                 const double* __restrict__ x_guess, const double* __restrict__ lam_guess,
                                                 const double* __restrict__ d, const double* __restrict__ lbx,
                                                 const double* __restrict__ ubx, const double* __restrict__ lbg,
                                                 const double* __restrict__ ubg, pmpc_sqp_settings ss, pmpc_qp_settings qs,
                                                 double* __restrict__ Hws, double* __restrict__ x, double* __restrict__ lam, pmpc_sqp_info* __restrict__ info,
                                                 double* __restrict__ slice_state, int* __restrict__ queue, unsigned lds_dyn_doubles, unsigned long long* __restrict__ prof) {
    static_assert(NN > 0 && NN + MM <= WAVE, "round-robin execution exists for the one-row-per-lane register path");
    extern __shared__ double smem[];
    const int P = cd->P, S = cd->S;
    Ocp<Model> ocp(model, P, S, cd->t_scale);
    const int n = ocp.dm.n, m = ocp.dm.m, mi = ocp.dm.mi;
    QpLds qw; SqpLds v;
    double* p = qw.carve_xy(smem, n, m);
    p = v.carve(p, n, m, mi);
    double* stage0 = p;
    p = ocp.s.carve(p, P, S);
    if ((size_t)(p - stage0) < (size_t)reg_qp_staging<NN + MM>() + 2 + ocp.s.const_doubles(P, S)) p = stage0 + reg_qp_staging<NN + MM>() + 2 + ocp.s.const_doubles(P, S);
    const double* stage_end = p;
    double* dL = p; p += (Model::ND > 0 ? Model::ND : 1);
    ocp.d = dL;
    const int ln = lane_id();
    ocp.stage_constants(cd);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc &= (unsigned)(RR_QUEUES - 1);
    const int nk = rr_queue_len(B, (int)xcc);
    if (nk == 0) return;
    const int cap = nk * ss.max_iter;
    int start = 0;
    for (int q = 0; q < (int)xcc; ++q) start += rr_queue_len(B, q) * ss.max_iter;
    int* head = queue + (size_t)xcc * 3 * RR_LINE;
    int* tail = head + RR_LINE;
    int* finished = head + 2 * RR_LINE;
    int* entries = queue + RR_HEADER_WORDS + start;
    const int G = WAVE / ocp.dm.NN;
    const bool side_by_side = (G >= 2 && (size_t)G * (m + ocp.dm.NN + 3) <= (size_t)(stage_end - ocp.s.fval));
#ifdef PMPC_RR_PROFILE   // developer build: cycles per wavefront in [0] whole life [1] solve [2] waiting for an entry [3] state load [4] state store + push; [5] work items [6] wavefronts
    long long pc[5] = {0, 0, 0, 0, 0}; int items = 0; const long long life0 = clock64();
#define RR_EXIT do { if (prof && ln == 0) { atomicAdd(&prof[0], (unsigned long long)(clock64() - life0)); for (int i_ = 1; i_ < 5; ++i_) atomicAdd(&prof[i_], (unsigned long long)pc[i_]); atomicAdd(&prof[5], (unsigned long long)items); atomicAdd(&prof[6], 1ull); } return; } while (0)
#define RR_T(var) const long long var = clock64()
#define RR_ACC(i, a, b) pc[i] += (b) - (a)
#else
#define RR_EXIT return
#define RR_T(var)
#define RR_ACC(i, a, b)
    (void)prof;
#endif
    long long cyc_sum = 0; int cyc_items = 0;   // this wavefront's iterations so far (shader-clock cycles)
    while (true) {
        RR_T(t0);
        int hq = 0;
        if (ln == 0) hq = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hq = __builtin_amdgcn_readfirstlane(hq);
        if (hq >= cap) RR_EXIT;   // beyond every push the queue can ever see
        int e = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&entries[hq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        for (int spin = 0; e == 0; ++spin) {
            if (spin >= RR_SPIN_LIMIT) RR_EXIT;   // (the resume-mode launch finishes what is left)
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= nk) RR_EXIT;
            __builtin_amdgcn_s_sleep(8);
            e = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&entries[hq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        const int b = (e & ((1 << RR_PASS_SHIFT) - 1)) - 1;
        const int pass = (int)((unsigned)e >> RR_PASS_SHIFT);   // SQP iterations the instance has behind it
        RR_T(t1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int i = ln; i < Model::ND; i += WAVE) dL[i] = d[(size_t)b * Model::ND + i];
        double* sst = slice_state + (size_t)b * 2 * n;   // [previous Lagrangian gradient | previous step]
        bool loose = false;   // the conditioning rule of the register-resident kernels (see sqp_kernel): unbounded controls / parameters -> the redo launch, before any work
        for (int i = ln; i < n; i += WAVE) {
            v.x[i] = (pass > 0) ? x[(size_t)b * n + i] : (x_guess ? x_guess[(size_t)b * n + i] : 0.0);
            const double lbi = lbx[(size_t)b * n + i], ubi = ubx[(size_t)b * n + i];
            v.lbx[i] = lbi; v.ubx[i] = ubi;
            loose |= (i >= ocp.dm.VARX) && (lbi < -LOOSE_BOUNDS_THRESH) && (ubi > LOOSE_BOUNDS_THRESH);
            if (pass > 0) { v.lg[i] = sst[i]; v.step[i] = sst[n + i]; }
        }
        const bool gave_up = __builtin_amdgcn_ballot_w64(loose) != 0;
        for (int i = ln; i < m + n; i += WAVE)
            v.lam[i] = (pass > 0) ? lam[(size_t)b * (m + n) + i] : (lam_guess ? lam_guess[(size_t)b * (m + n) + i] : 0.0);
        for (int i = ln; i < mi; i += WAVE) {
            v.lbg[i] = lbg ? lbg[(size_t)b * mi + i] : -INFINITY;
            v.ubg[i] = ubg ? ubg[(size_t)b * mi + i] : INFINITY;
        }
        wsync();
        RR_T(t2);
        double* K0 = Hws + (size_t)b * (size_t)(n + m) * n;
        pmpc_sqp_info si;
        int done_iters = pass;
        if (gave_up) { si.iter = 0; si.qp_solver_iter = 0; si.status = PMPC_SQP_REDO; si.flags = PMPC_FLAG_ILLCOND; si.primal_norm = 0.0; si.dual_norm = 0.0; si.max_violation = 0.0; si.cost = 0.0; }
        else {
            SqpDevice<Model, NN, MM, false, HU, false> sqp(ocp, v, qw, K0, K0 + n, ss, qs);
            sqp.filt = nullptr;
            sqp.eig = nullptr;
            sqp.trace = ss.iteration_trace ? ss.iteration_trace + (size_t)b * (size_t)ss.iteration_trace_capacity * PMPC_TRACE_DOUBLES : nullptr;
            sqp.tr = ocp.s.fval;
            sqp.lsbuf = ocp.s.fval;
            sqp.ls_side_by_side = side_by_side;
            if (pass > 0) { const pmpc_sqp_info prev = info[b]; sqp.qp_iter_total = prev.qp_solver_iter; sqp.qp_flags = prev.flags; sqp.cost_log = prev.cost; }
            // An instance whose iteration took much longer than this wavefront's iterations so far (hard QPs: more ADMM iterations, rho updates with their
            // re-factorisations) is on the critical path of the launch: it keeps the wavefront until it ends instead of queueing behind the others.
            while (true) {
                const long long c0 = clock64();
                sqp.solve(si, done_iters, done_iters + 1);
                const long long c1 = clock64() - c0;
                ++done_iters;
                if (si.status != PMPC_SQP_IN_PROGRESS) break;
                const bool hard = cyc_items > 0 && c1 * 10 * cyc_items > cyc_sum * RR_HARD_TENTHS;
                cyc_sum += c1; ++cyc_items;
                if (!hard) break;
            }
        }
        RR_T(t3);
#ifdef PMPC_RR_PROFILE   // per-item wall-clock stamps (100 MHz) in the alpha / primal_norm / dual_norm fields of the iteration record
        if (ss.iteration_trace && ln == 0 && si.iter <= ss.iteration_trace_capacity) {
            double* r_ = ss.iteration_trace + ((size_t)b * (size_t)ss.iteration_trace_capacity + (size_t)(si.iter - 1)) * PMPC_TRACE_DOUBLES;
            r_[1] = (double)wall_clock64(); r_[2] = (double)(t2 - t0); r_[3] = (double)(t3 - t2); r_[4] = (double)xcc;
        }
#endif
        const bool more = si.status == PMPC_SQP_IN_PROGRESS;
        int tq = 0;
        if (more && ln == 0) tq = __hip_atomic_fetch_add(tail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (reserved while the stores below are in flight)
        if (more) for (int i = ln; i < n; i += WAVE) { sst[i] = v.lg[i]; sst[n + i] = v.step[i]; }
        for (int i = ln; i < n; i += WAVE) x[(size_t)b * n + i] = v.x[i];
        for (int i = ln; i < m + n; i += WAVE) lam[(size_t)b * (m + n) + i] = v.lam[i];
        if (ln == 0) info[b] = si;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's stores have reached the XCD's L2
        if (ln == 0) {
            if (more) { if (tq < cap) __hip_atomic_store(&entries[tq], (b + 1) | (done_iters << RR_PASS_SHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else __hip_atomic_fetch_add(finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wsync();
#ifdef PMPC_RR_PROFILE
        { const long long t4 = clock64(); RR_ACC(2, t0, t1); RR_ACC(3, t1, t2); RR_ACC(1, t2, t3); RR_ACC(4, t3, t4); ++items; }
#endif
    }
#undef RR_EXIT
#undef RR_T
#undef RR_ACC
}
// mode 0: KKT factor in LDS; 1: register-resident QP (n+m <= 64); 2: KKT factor in HBM (large instances); 3: register-resident QP, 65..112 rows
template <class Model> inline size_t sqp_eig_lds_bytes(int P, int S, const pmpc_sqp_settings* ss) {   // regularisation = 1: Jacobi workspace
    OcpDims<Model> dm(P, S);
    return ss->regularisation == 1 ? 2 * (size_t)dm.n * dm.n * sizeof(double) : 0;
}
template <class Model> inline size_t sqp_kernel_lds_bytes(int P, int S, int mode, int qp_solver = 0, bool pol = false, size_t cnd_staging = 0) {   // mode 6: condensed register QP on more than 64 variables, staging = cnd_staging (cond_qp_staging)
    OcpDims<Model> dm(P, S);
    if (mode == 0 && qp_solver == 1)
        return (QpLds::doubles(dm.n, dm.m + dm.n) + SqpLds::doubles(dm.n, dm.m, dm.mi) + OcpLds<Model>::doubles(P, S) + 8 + FILTER_LDS_DOUBLES) * sizeof(double);
    size_t stage = OcpLds<Model>::doubles(P, S);
    if (mode == 1) { const size_t need = (size_t)RegKkt<64>::TRI + OcpLds<Model>::const_doubles(P, S) + 8; if (stage < need) stage = need; }
    if (mode == 2)   // large instances: x, y, the right-hand side of the substitutions and the factorisation's diagonal tile; everything else in HBM
        return (QpLds::doubles_xy(dm.n, dm.m) + (size_t)(dm.n + dm.m) + BigKkt::LDS_DOUBLES + BIG_MAIL_DOUBLES + (size_t)(P + 1) * (P + 2) + ((Model::NG == 0 && Model::NP == 0 && JViewRT<Model>::tab_worth_it(dm.NN)) ? JViewRT<Model>::tab_doubles(dm.NN) + 1 : 0) + (Model::ND > 0 ? Model::ND : 1) + FILTER_LDS_DOUBLES + 8) * sizeof(double);
    if (mode == 3) { const size_t need = (size_t)RegKkt2<112>::TRI + OcpLds<Model>::const_doubles(P, S) + 8; if (stage < need) stage = need; }
    if (mode == 5) { const size_t need = (size_t)RegKkt<64>::TRI + OcpLds<Model>::const_doubles(P, S) + 8; if (stage < need) stage = need; }   // condensed register QP on at most 64 variables (65..128 KKT rows)
    if (mode == 4) { const size_t need = (size_t)RegKkt2<128>::TRI + OcpLds<Model>::const_doubles(P, S) + 8; if (stage < need) stage = need; }   // 113..128 rows: LDS-resident operand tiles
    if (mode == 6) { const size_t need = cnd_staging + OcpLds<Model>::const_doubles(P, S) + 8; if (stage < need) stage = need; }
    return ((mode == 0 ? QpLds::doubles(dm.n, dm.m) : QpLds::doubles_xy(dm.n, dm.m)) + SqpLds::doubles(dm.n, dm.m, dm.mi) + stage + 8 +
            ((mode == 1 || mode == 3 || mode == 4 || mode == 5 || mode == 6) ? (mode == 1 ? 0 : jview_doubles<Model>(dm.NN)) + (pol ? FILTER_LDS_DOUBLES : 0) : FILTER_LDS_DOUBLES) + (mode == 2 ? BigKkt::LDS_DOUBLES : 0)) * sizeof(double);
}
constexpr int BIG_WG4_MAX_BATCH = 256;   // instances (on a 256-CU device) up to which the four-wavefront team kernel serves a large-instance batch (see sqp_launch_dev; measured: one workgroup per CU — 256: 11.9 -> 9.9 ms, 512: 13.5 -> 19.2)
constexpr int BIG_TWO_WAVES_MAX_ROWS = 200;   // below: two wavefronts per SIMD on the HBM-factor kernel when the batch exceeds the SIMD count (see sqp_launch_dev)
constexpr int BIG_KKT_MIN_ROWS = 96;   // n + m from which sqp_launch_dev prefers the HBM-factor kernel (see there)
template <class Model> inline bool sqp_hbm_mode_fits(int P, int S) {   // do the QP vectors fit the second-order staging?
    OcpDims<Model> dm(P, S);
    return QpLds::doubles_rest(dm.n, dm.m) <= 2 * (size_t)dm.NN * OcpDims<Model>::NDER * OcpDims<Model>::NDER;
}

// collocation assembly only (used to check A2/A4/A6/A7/A8/A9/A10 against the reference's golden vectors)
template <class Model>
__global__ __launch_bounds__(64) void linearise_kernel(Model model, const ChebData* __restrict__ cd, int B,
                                                       const double* __restrict__ var, const double* __restrict__ d,
                                                       const double* __restrict__ lam, double* __restrict__ cost,
                                                       double* __restrict__ constr, double* __restrict__ jac,
                                                       double* __restrict__ cost_grad, double* __restrict__ lag_grad,
                                                       double* __restrict__ lag_hess) {
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int P = cd->P, S = cd->S;
    Ocp<Model> ocp(model, P, S, cd->t_scale);
    const int n = ocp.dm.n, m = ocp.dm.m;
    double* p = ocp.s.carve(smem, P, S);
    double* xL = p; p += n; double* lamL = p; p += m + n; double* cL = p; p += m; double* gL = p; p += n;
    double* dL = p; p += (Model::ND > 0 ? Model::ND : 1);
    const int ln = lane_id();
    for (int i = ln; i < Model::ND; i += WAVE) dL[i] = d[(size_t)b * Model::ND + i];
    ocp.d = dL;
    ocp.stage_constants(cd);
    for (int i = ln; i < n; i += WAVE) xL[i] = var[(size_t)b * n + i];
    for (int i = ln; i < m + n; i += WAVE) lamL[i] = lam ? lam[(size_t)b * (m + n) + i] : 0.0;
    wsync();
    ocp.stage_first_order(xL);
    ocp.stage_second_order(xL, lamL);
    double* J = jac + (size_t)b * m * n;
    double* Hh = lag_hess + (size_t)b * n * n;
    const double cst = ocp.assemble_first_order(cL, J, gL, m);
    ocp.assemble_hessian(Hh, n);
    for (int j = ln; j < n; j += WAVE) {
        double a = 0.0;
        for (int i = 0; i < m; ++i) a += J[(size_t)j * m + i] * lamL[i];
        a += gL[j];
        a += lamL[m + j];
        lag_grad[(size_t)b * n + j] = a;
        cost_grad[(size_t)b * n + j] = gL[j];
    }
    // values-only paths (cost / constraints), as the line search uses them
    const double cst2 = ocp.cost(xL);
    ocp.constraints(xL, cL);
    for (int i = ln; i < m; i += WAVE) constr[(size_t)b * m + i] = cL[i];
    if (ln == 0) { cost[2 * b] = cst; cost[2 * b + 1] = cst2; }
}
template <class Model> inline size_t linearise_kernel_lds_bytes(int P, int S) {
    OcpDims<Model> dm(P, S);
    return (OcpLds<Model>::doubles(P, S) + 3 * (size_t)dm.n + 2 * (size_t)dm.m + 16) * sizeof(double);
}


// models whose LDS-resident kernel also exists with phase timers (PMPC_PHASE_PROFILE=1)
template <class Model> struct LDS_PATH_PROFILED { static constexpr bool value = false; };
// (specialised HERE, before any template reads it and in the one header every translation unit sees: a specialisation that only some translation
// units declare would make the trait's value depend on the translation unit — an ODR violation)
template <> struct LDS_PATH_PROFILED<RobotOCP> { static constexpr bool value = true; };
template <> struct LDS_PATH_PROFILED<CstrOCP> { static constexpr bool value = true; };
template <> struct LDS_PATH_PROFILED<KiteStandInOCP> { static constexpr bool value = true; };

// Launch the fused SQP kernel for `Model` on DEVICE buffers (asynchronous on the context's stream).
#ifndef PMPC_EXPERIMENT_COND_SMALL
#define PMPC_EXPERIMENT_COND_SMALL 0
#endif
#ifndef PMPC_EXPERIMENT_SMALL_POL
#define PMPC_EXPERIMENT_SMALL_POL 0   /* 2: developer switch — the hook build of the small condensed kernel also under the DEFAULT policies (the bisection of EXPERIMENTS.md round 5, with -DPMPC_EXPERIMENT_CND_WITH_RUIZ) */
#endif
// Redo launch behind a one-row-per-lane register kernel (PMPC_FLAG_ILLCOND, include/polympc_amd.h): the instances whose constraint-first QP gave up at
// its conditioning gate (none on any BASELINE workload) are solved again, from their guesses, by the LDS-resident kernel — static LDL^T of the
// (n + m)-row KKT matrix with substitutions; every other workgroup reads one word and exits. PMPC_NO_REDO_LAUNCH=1: developer switch (timing the launch).
template <class Model>
inline bool launch_redo_generic(pmpc_context* ctx, const Model& mdl, const ChebData* cd, int P, int S, int B, const double* x_guess, const double* lam_guess, const double* d,
                                const double* lbx, const double* ubx, const double* lbg, const double* ubg, const pmpc_sqp_settings* ss,
                                const pmpc_qp_settings* qs, double* Hws, double* Aws, double* x, double* lam, pmpc_sqp_info* info, hipStream_t stream,
                                size_t lds_limit) {
    if (pmpc_internal_switch(ctx, PMPC_SW_NO_REDO_LAUNCH)) return true;
    const size_t ldsg = sqp_kernel_lds_bytes<Model>(P, S, 0, 0) + sqp_eig_lds_bytes<Model>(P, S, ss);
    if (ldsg > lds_limit) return true;   // (systems of at most 64 rows always fit)
    auto gk = sqp_kernel<Model>;
    if (hipFuncSetAttribute((const void*)gk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg) != hipSuccess) return false;
    pmpc_sqp_settings ssf = *ss; ssf.kkt_form = 1;
    hipLaunchKernelGGL(gk, dim3(B), dim3(WAVE), ldsg, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ssf, *qs, Hws, Aws, x, lam, info,
                       (unsigned long long*)nullptr, (double*)nullptr, PMPC_REDO_MODE, ss->max_iter, (double*)nullptr, (unsigned)(ldsg / sizeof(double)));
    return true;
}
template <class Model, int NN_, int MM_> struct COND_REG_OK { static constexpr bool value = NN_ + MM_ > WAVE && NN_ <= 112 && MM_ > 0 && MM_ <= WAVE && Model::NP <= 1; };   // (NP = 1 and NG > 0 since round 6: the parameter's dense column of A as a wave reduction, the path-constraint rows as own-node blocks without a D~ row)
// Register-resident QP specialisations are selected from the compile-time model dimensions and the runtime node count
// when the KKT system has at most 64 rows; otherwise the LDS-resident path is used.
template <class Model, int NNODES, bool LEAN = false>   // LEAN: no phase-timer and no block-BFGS specialisation (pmpc_grids.hpp: those requests take the LDS-resident kernel)
inline bool try_launch_reg(pmpc_context* ctx, const Model& mdl, const ChebData* cd, int P, int S, int B, const double* x_guess,
                           const double* lam_guess, const double* d, const double* lbx, const double* ubx, const double* lbg,
                           const double* ubg, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* Hws, double* Aws, double* x,
                           double* lam, pmpc_sqp_info* info, hipStream_t stream, size_t lds_limit, unsigned long long* phase, pmpc_status* st,
                           double* slice_state, int slice_iters) {
    constexpr int NN_ = (Model::NX + Model::NU) * NNODES + Model::NP;
    constexpr int MM_ = (Model::NX + Model::NG) * NNODES;
    // the policy hooks the reference's tests install beside the defaults — Ruiz preconditioner, filter line search — exist on the register paths
    // for the grids of its own tests (7, 11 and — where the model fits 128 rows — 16 nodes) as separate kernels (POL); any other grid takes the LDS /
    // HBM-resident kernels for them
    const bool pol = ss->preconditioner == 1 || ss->line_search == 1 || ss->regularisation == 1;   // (regularisation = 1, eigenvalue mirroring — sqp_test_autodiff.cpp:29-45: Jacobi workspace of 16 n^2 bytes of LDS, round 6)
    constexpr bool POLK = ((!LEAN && (NNODES == 7 || NNODES == 11)) || NNODES == 16) && (int)OcpDims<Model>::NDER <= RUIZ_MAX_NDER;   // (16 nodes: the reference's mpc_wrapper_test grid, round 4)
    if (pol && !POLK) return false;
    if constexpr (NN_ + MM_ <= WAVE) {
        if (P * S + 1 != NNODES) return false;
        const size_t ldsr = sqp_kernel_lds_bytes<Model>(P, S, 1, 0, pol) + sqp_eig_lds_bytes<Model>(P, S, ss);
        if (ldsr > lds_limit) return false;
        if (LEAN && (ss->hessian_update == 1 || phase)) return false;
        pmpc_internal_set_route(ctx, PMPC_ROUTE_REG1);
        auto kern = sqp_kernel<Model, NN_, MM_, false>;
        if constexpr (!LEAN)
            kern = (ss->hessian_update == 1) ? sqp_kernel<Model, NN_, MM_, false, 1>   // (no phase timers in the block-BFGS specialisation)
                                             : (phase ? sqp_kernel<Model, NN_, MM_, true> : sqp_kernel<Model, NN_, MM_, false>);
        if constexpr (POLK) { if (pol) kern = (ss->hessian_update == 1) ? sqp_kernel<Model, NN_, MM_, false, 1, false, false, true> : sqp_kernel<Model, NN_, MM_, false, 0, false, false, true>; }
#if PMPC_EXPERIMENT_COND_SMALL   /* developer experiment (round 6, EXPERIMENTS.md): the CONDENSED register QP (pmpc_qp_cond.hpp) on a grid of at most 64 KKT rows — config A on 35 instead of 56 rows */
        size_t ldsc = ldsr;
        if constexpr (!LEAN && NNODES == 7 && Model::NP == 0 && Model::NG == 0) {
            if (!pol && ss->kkt_form == 0 && ss->hessian_update == 0 && !phase && !pmpc_internal_switch(ctx, PMPC_SW_NO_CONDREG)) {
                kern = sqp_kernel<Model, NN_, MM_, false, 0, false, false, false, true>;
                ldsc = sqp_kernel_lds_bytes<Model>(P, S, 5, 0, false);
                pmpc_internal_set_route(ctx, PMPC_ROUTE_CONDREG);
            }
        }
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc) != hipSuccess) { *st = PMPC_ERR_HIP; return true; }
        if (pmpc_internal_last_route(ctx) == PMPC_ROUTE_CONDREG) {
            hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), ldsc, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg,
                               *ss, *qs, Hws, Aws, x, lam, info, phase, (double*)nullptr, 0, ss->max_iter, (double*)nullptr, (unsigned)(ldsc / sizeof(double)));
            if (!launch_redo_generic<Model>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, lds_limit)) { *st = PMPC_ERR_HIP; return true; }
            *st = (hipGetLastError() == hipSuccess) ? PMPC_OK : PMPC_ERR_HIP;
            return true;
        }
#endif
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr) != hipSuccess) { *st = PMPC_ERR_HIP; return true; }
        const int slice = (slice_state && slice_iters > 0) ? slice_iters : ss->max_iter;
        if constexpr (!LEAN) {
            // more instances than resident wavefronts: one SQP iteration per work item, round-robin over the instances (sqp_kernel_rr)
            const int slots = PMPC_SQP_WAVES * pmpc_internal_simd_count(ctx);
            #ifdef PMPC_RR_PROFILE
            const bool rr_phase_ok = true;
#else
            const bool rr_phase_ok = !phase;
#endif
            if (!pol && slice_state && slice_iters == 0 && rr_phase_ok && ss->max_iter > 1 && ss->max_iter <= RR_MAX_ITER && B > slots && B <= RR_MAX_BATCH && (size_t)B * ss->max_iter < ((size_t)1 << 30) && pmpc_internal_sqp_rr(ctx)) {
                auto rrk = (ss->hessian_update == 1) ? sqp_kernel_rr<Model, NN_, MM_, 1> : sqp_kernel_rr<Model, NN_, MM_, 0>;
                if (hipFuncSetAttribute((const void*)rrk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr) != hipSuccess) { *st = PMPC_ERR_HIP; return true; }
                int* queue = (int*)(slice_state + (size_t)B * 2 * NN_);   // behind the slice state (sqp_launch_dev sizes the workspace for it)
                { const size_t words = (size_t)B * ss->max_iter > (size_t)RR_HEADER_WORDS ? (size_t)B * ss->max_iter : (size_t)RR_HEADER_WORDS;
                  hipLaunchKernelGGL(sqp_rr_init_kernel<Model>, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, queue, info, B, ss->max_iter); }
                hipLaunchKernelGGL(rrk, dim3(slots), dim3(WAVE), ldsr, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg,
                                   *ss, *qs, Hws, x, lam, info, slice_state, queue, (unsigned)(ldsr / sizeof(double)), phase);
                // resume mode: a no-op per instance unless something was left in progress (see sqp_kernel_rr)
                hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), ldsr, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg,
                                   *ss, *qs, Hws, Aws, x, lam, info, phase, (double*)nullptr, -1, ss->max_iter, slice_state, (unsigned)(ldsr / sizeof(double)));
                if (!launch_redo_generic<Model>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, lds_limit)) { *st = PMPC_ERR_HIP; return true; }
                *st = (hipGetLastError() == hipSuccess) ? PMPC_OK : PMPC_ERR_HIP;
                return true;
            }
        }
        for (int it = 0; it < ss->max_iter; it += slice)
            hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), ldsr, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg,
                               *ss, *qs, Hws, Aws, x, lam, info, phase, (double*)nullptr, it, it + slice, slice_state, (unsigned)(ldsr / sizeof(double)));
        if (!launch_redo_generic<Model>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, lds_limit)) { *st = PMPC_ERR_HIP; return true; }
        *st = (hipGetLastError() == hipSuccess) ? PMPC_OK : PMPC_ERR_HIP;
        return true;
    } else if constexpr (NN_ + MM_ <= 128) {   // two KKT rows per lane (pmpc_qp_reg2.hpp); the Hessian-update policy is a run-time choice there
        if (P * S + 1 != NNODES) return false;
        const size_t ldsr = sqp_kernel_lds_bytes<Model>(P, S, NN_ + MM_ <= 112 ? 3 : 4, 0, pol) + sqp_eig_lds_bytes<Model>(P, S, ss);
        if (ldsr > lds_limit) return false;
        pmpc_internal_set_route(ctx, PMPC_ROUTE_REG2);
        auto kern = sqp_kernel<Model, NN_, MM_, false>;
        bool timed = false;
        if constexpr (LDS_PATH_PROFILED<Model>::value && !LEAN) { if (phase && !pol) { kern = sqp_kernel<Model, NN_, MM_, true>; timed = true; } }   // developer builds with phase timers
        if constexpr (POLK) { if (pol) kern = sqp_kernel<Model, NN_, MM_, false, 0, false, false, true>; }
        // condensed register QP (pmpc_qp_cond.hpp): 65..112 variables, at most 64 constraint rows, the default policies; kkt_form = 1 keeps the full inverse
        size_t ldsq = ldsr;
        if constexpr (COND_REG_OK<Model, NN_, MM_>::value) {
            if (!pol && ss->kkt_form == 0 && !pmpc_internal_switch(ctx, PMPC_SW_NO_CONDREG)) {
                if constexpr (NN_ <= WAVE) ldsq = sqp_kernel_lds_bytes<Model>(P, S, 5, 0, false);   // (two wavefronts per SIMD: a smaller staging lets more instances share a CU)
                else ldsq = sqp_kernel_lds_bytes<Model>(P, S, 6, 0, false, (size_t)cond_qp_staging<NN_, MM_, NNODES>());   // (its own tile set's staging, not the full inverse's: four instead of three instances per CU on the 16-node grid)
                kern = sqp_kernel<Model, NN_, MM_, false, 0, false, false, false, true>; timed = false;
                if constexpr (LDS_PATH_PROFILED<Model>::value && !LEAN) { if (phase) { kern = sqp_kernel<Model, NN_, MM_, true, 0, false, false, false, true>; timed = true; } }
                pmpc_internal_set_route(ctx, PMPC_ROUTE_CONDREG);
#if PMPC_EXPERIMENT_SMALL_POL == 2   /* developer experiment: the hook variant of the small condensed kernel under the DEFAULT policies */
                if constexpr (POLK && NN_ <= WAVE) { kern = sqp_kernel<Model, NN_, MM_, false, 0, false, false, true, true>; ldsq = sqp_kernel_lds_bytes<Model>(P, S, 3, 0, true); }
#endif
            }
            // the hooks keep the condensed QP — the filter line search and eigenvalue mirroring since rounds 4 / 6, the Ruiz preconditioner since late round 6: it rescales the
            // workspace, so the hook builds read their D~ tables (one set per state index) and, after an equilibration, the node blocks back from it (pmpc_qp_cond.hpp WS)
            if constexpr (POLK) {
                if (pol && ss->kkt_form == 0 && !pmpc_internal_switch(ctx, PMPC_SW_NO_CONDREG) && !(ss->preconditioner == 1 && pmpc_internal_switch(ctx, PMPC_SW_NO_CONDREG_RUIZ))) {
                    const size_t stg = ss->preconditioner == 1 ? (size_t)cond_qp_staging_ws<NN_, MM_, NNODES, Model::NX>() : (size_t)cond_qp_staging<NN_, MM_, NNODES>();   // (the tables per state index only with the Ruiz preconditioner: the other hooks keep the smaller staging — more instances per CU)
                    const size_t ldsw = sqp_kernel_lds_bytes<Model>(P, S, 6, 0, true, stg) + sqp_eig_lds_bytes<Model>(P, S, ss);
                    if (ldsw <= lds_limit) {
                        kern = sqp_kernel<Model, NN_, MM_, false, 0, false, false, true, true>; timed = false;
                        ldsq = ldsw;
                        pmpc_internal_set_route(ctx, PMPC_ROUTE_CONDREG);
                    }
                }
            }
        }
        auto kern_full = sqp_kernel<Model, NN_, MM_, false>;   // the full two-rows-per-lane inverse: serves the redo launch behind a condensed kernel
        if constexpr (POLK) { if (pol) kern_full = sqp_kernel<Model, NN_, MM_, false, 0, false, false, true>; }
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq) != hipSuccess) { *st = PMPC_ERR_HIP; return true; }
        const int slice = (slice_state && slice_iters > 0) ? slice_iters : ss->max_iter;
        for (int it = 0; it < ss->max_iter; it += slice)
            hipLaunchKernelGGL(kern, dim3(B), dim3(WAVE), ldsq, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg,
                               *ss, *qs, Hws, Aws, x, lam, info, timed ? phase : (unsigned long long*)nullptr, (double*)nullptr, it, it + slice, slice_state, (unsigned)(ldsq / sizeof(double)));
        if (pmpc_internal_last_route(ctx) == PMPC_ROUTE_CONDREG && !pmpc_internal_switch(ctx, PMPC_SW_NO_REDO_LAUNCH)) {
            // redo launch: the instances whose condensed solve gave up at its conditioning gate (PMPC_FLAG_ILLCOND; none on any BASELINE workload) are solved
            // again, from their guesses, by the full-inverse kernel of this size — every other workgroup reads one word and exits
            if (hipFuncSetAttribute((const void*)kern_full, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr) != hipSuccess) { *st = PMPC_ERR_HIP; return true; }
            hipLaunchKernelGGL(kern_full, dim3(B), dim3(WAVE), ldsr, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg,
                               *ss, *qs, Hws, Aws, x, lam, info, (unsigned long long*)nullptr, (double*)nullptr, PMPC_REDO_MODE, ss->max_iter, (double*)nullptr, (unsigned)(ldsr / sizeof(double)));
        }
        *st = (hipGetLastError() == hipSuccess) ? PMPC_OK : PMPC_ERR_HIP;
        return true;
    } else {
        return false;
    }
}

// Register-resident specialisations for further node counts live in their own translation units (pmpc_grids_*.hip, built-in models only: a user OCP's
// translation unit instantiates the three grids below and takes the LDS / HBM-factor kernels elsewhere). Same arguments and meaning as try_launch_reg.
template <class Model> struct EXTRA_GRIDS { static constexpr bool value = false; };
template <> struct EXTRA_GRIDS<RobotOCP> { static constexpr bool value = true; };      // (declared here for every translation unit, see LDS_PATH_PROFILED)
template <> struct EXTRA_GRIDS<CstrOCP> { static constexpr bool value = true; };
template <> struct EXTRA_GRIDS<ParkingOCP> { static constexpr bool value = true; };
template <> struct EXTRA_GRIDS<RobotNGOCP> { static constexpr bool value = true; };
template <> struct EXTRA_GRIDS<ParkingNGOCP> { static constexpr bool value = true; };
template <class Model>
bool try_launch_extra_grids(pmpc_context* ctx, const Model& mdl, const ChebData* cd, int P, int S, int B, const double* x_guess,
                            const double* lam_guess, const double* d, const double* lbx, const double* ubx, const double* lbg,
                            const double* ubg, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* Hws, double* Aws, double* x,
                            double* lam, pmpc_sqp_info* info, hipStream_t stream, size_t lds_limit, unsigned long long* phase, pmpc_status* st,
                            double* slice_state, int slice_iters);

template <class Model>
inline pmpc_status sqp_launch_dev(pmpc_context* ctx, const Model& mdl, int P, int S, double t0, double tf, int B, const double* x_guess,
                                  const double* lam_guess, const double* d, const double* lbx, const double* ubx, const double* lbg,
                                  const double* ubg, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam,
                                  pmpc_sqp_info* info) {
    if (P < 1 || P > MAX_P || S < 1 || P * S + 1 > MAX_NODES) return PMPC_ERR_UNSUPPORTED_SIZE;
    if (ss->preconditioner != 0 && (ss->preconditioner != 1 || (int)OcpDims<Model>::NDER > RUIZ_MAX_NDER)) return PMPC_ERR_UNSUPPORTED_SIZE;
    OcpDims<Model> dm(P, S);
    const void* cdv = nullptr; double* ws = nullptr; void* streamv = nullptr; size_t lds_limit = 0; unsigned long long* phase = nullptr; int force_lds = 0;
    const size_t base = (size_t)B * ((size_t)dm.n * dm.n + (size_t)dm.m * dm.n + 2 * (size_t)dm.n);
    // the ready queue of the round-robin kernel (developer switch PMPC_SQP_RR=1, one-row-per-lane register path only) is requested only when that
    // kernel can run: with the reference's default max_iter = 100 it is B x 400 bytes of otherwise dead HBM
    const bool rr_possible = pmpc_internal_sqp_rr(ctx) && dm.n + dm.m <= WAVE && ss->max_iter > 1 && ss->max_iter <= RR_MAX_ITER && B <= RR_MAX_BATCH &&
                             (size_t)B * ss->max_iter < ((size_t)1 << 30);
    pmpc_status st = pmpc_internal_services(ctx, P, S, t0, tf, base * sizeof(double) + (rr_possible ? rr_queue_bytes(B, ss->max_iter) : 0), &cdv, &ws, &streamv, &lds_limit, &phase, &force_lds);
    if (st != PMPC_OK) return st;
    const ChebData* cd = (const ChebData*)cdv;
    hipStream_t stream = (hipStream_t)streamv;
    double* Hws = ws; double* Aws = ws + (size_t)B * dm.n * dm.n;
    double* slice_state = Aws + (size_t)B * dm.m * dm.n;
    const int slice_iters = ss->line_search == 1 ? 0 : pmpc_internal_sqp_slice(ctx);   // (the filter lives in LDS for the whole solve)
    if ((ss->hessian_update != 0 && ss->hessian_update != 1) || (ss->qp_solver != 0 && ss->qp_solver != 1)) return PMPC_ERR_INVALID_ARGUMENT;
    if ((ss->line_search != 0 && ss->line_search != 1) ||
        (ss->line_search == 1 && (ss->filter_max_depth < 1 || ss->filter_max_depth > PMPC_FILTER_MAX_DEPTH))) return PMPC_ERR_INVALID_ARGUMENT;
    if (qs->linear_solver != 0 && qs->linear_solver != 1) return PMPC_ERR_INVALID_ARGUMENT;
    if constexpr (SCHUR_GRIDS<Model>::value) {   // block-diagonal Hessian on a grid with a block-structured kernel
        // (the redo launch behind this kernel's conditioning gate runs the LDS-resident kernel: a grid it does not fit — only under a developer's PMPC_LDS_LIMIT —
        //  keeps the dense kernels, so that an instance that gave up can always be solved again)
        if (!force_lds && !pmpc_internal_switch(ctx, PMPC_SW_NO_SCHUR) && schur_request_ok(ss, qs, slice_iters) && sqp_kernel_lds_bytes<Model>(P, S, 0, 0) <= lds_limit) {
            pmpc_status rst = PMPC_OK;
            if (try_launch_schur_grids<Model>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, ss, qs, x, lam, info, stream, lds_limit, phase, &rst)) {
                // redo launch: the instances whose block-structured QP gave up at its conditioning gate (PMPC_SCHUR_COND_GATE; none on any BASELINE workload) are
                // solved again, from their guesses, on the LDS-resident static LDL^T of the (n + m)-row matrix (every compiled grid fits)
                if (rst == PMPC_OK && !launch_redo_generic<Model>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, lds_limit)) rst = PMPC_ERR_HIP;
                return rst;
            }
        }
    }
    // kkt_form = 2 asks for the block-structured range-space form where a specialisation is compiled (above); everywhere else it means the default
    pmpc_sqp_settings ss_default_form;
    if (ss->kkt_form == 2) { ss_default_form = *ss; ss_default_form.kkt_form = 0; ss = &ss_default_form; }
    if (!force_lds && ss->qp_solver == 0 && qs->linear_solver == 0) {   // (preconditioner / line_search = 1: the 7- and 11-node register kernels carry them, see try_launch_reg)   // node counts whose KKT system can fit 64 rows for small models (7 nodes: config A / D); Ruiz: LDS path
        pmpc_status rst = PMPC_OK;
        if (try_launch_reg<Model, 7>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, lds_limit, phase, &rst, slice_state, slice_iters)) return rst;
        if (try_launch_reg<Model, 5>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, lds_limit, phase, &rst, slice_state, slice_iters)) return rst;
        if (try_launch_reg<Model, 11>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, lds_limit, phase, &rst, slice_state, slice_iters)) return rst;   // config B / the reference's P = 5, S = 2 grids: 88..110 KKT rows
        if constexpr (EXTRA_GRIDS<Model>::value) {   // 3-, 4-, 6-, 8-, 9-, 10-, 12- and 13-node grids of the built-in models
            if (try_launch_extra_grids<Model>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, lds_limit, phase, &rst, slice_state, slice_iters)) return rst;
        }
    }
    size_t lds = sqp_kernel_lds_bytes<Model>(P, S, 0, ss->qp_solver) + sqp_eig_lds_bytes<Model>(P, S, ss);
    double* Kws = nullptr;
    if (lds > lds_limit && ss->qp_solver == 1) return PMPC_ERR_UNSUPPORTED_SIZE;   // the stacked system lives in LDS only
    if (lds > lds_limit && qs->linear_solver == 1) return PMPC_ERR_UNSUPPORTED_SIZE;   // the pivoted factorisation lives in LDS only
    // Above BIG_KKT_MIN_ROWS rows the blocked tile factorisation (fp64 MFMA trailing updates, factor panels streamed from HBM / L2, 13 KB of LDS:
    // one instance per SIMD) beats the LDS-resident triangle, whose footprint leaves one or two instances per CU — measured on robot grids, 2048
    // instances: 104 rows 21.0 -> 18.3 ms, 128 rows (the reference's mpc_wrapper_test grid) 52.2 -> 25.3, 168 rows 247.6 -> 39.4; 72 rows 13.5 -> 17.7
    // (stays in LDS). The stacked OSQP-form system and the pivoted factorisation exist in LDS only.
    const bool prefer_big = dm.n + dm.m >= BIG_KKT_MIN_ROWS && ss->qp_solver == 0 && qs->linear_solver == 0 && !force_lds;
    if (lds > lds_limit || prefer_big) {   // large instance: KKT factor in HBM, SQP / QP vectors in an HBM scratch behind it
        lds = sqp_kernel_lds_bytes<Model>(P, S, 2) + sqp_eig_lds_bytes<Model>(P, S, ss);
        if (lds > lds_limit) return PMPC_ERR_UNSUPPORTED_SIZE;
        const size_t base16 = (base + 1) & ~(size_t)1;   // (16-byte boundary for the factor workspaces)
        st = pmpc_internal_services(ctx, P, S, t0, tf, (base16 + (size_t)B * (BigKkt::doubles(dm.n + dm.m) + big_scratch_doubles<Model>(P, S))) * sizeof(double), &cdv, &ws, &streamv,
                                    &lds_limit, &phase, &force_lds);
        if (st != PMPC_OK) return st;
        Hws = ws; Aws = ws + (size_t)B * dm.n * dm.n; slice_state = Aws + (size_t)B * dm.m * dm.n; Kws = ws + base16;
    }
    pmpc_internal_set_route(ctx, Kws ? PMPC_ROUTE_HBM : PMPC_ROUTE_LDS);
    auto lkern = Kws ? sqp_kernel<Model, 0, 0, false, 0, true> : sqp_kernel<Model>;
    // Mid-size instances on the HBM-factor kernel stream little per wavefront: with more instances than SIMDs a second wavefront per SIMD (a 256-register
    // build of the same kernel) hides part of that latency — per 4096 robots 128 rows 40.5 -> 35.4 ms, 168 rows 64.3 -> 60.8; at config C's 464 rows
    // the second wave only thrashes the L2 (2048 instances 93.0 -> 99.1 ms), hence the row bound.
    if (Kws && dm.n + dm.m < BIG_TWO_WAVES_MAX_ROWS && B > pmpc_internal_simd_count(ctx) && !phase) lkern = sqp_kernel<Model, 0, 0, false, 0, true, true>;
    // Small batches of large instances: a workgroup of four wavefronts per instance (BigTeam, pmpc_qp_big.hpp). With at most one instance per CU the chip is a
    // quarter full at best on the one-wavefront kernel and the launch takes a lone instance's time; the team cuts that time (the condensed build, the blocked
    // factorisation and the two triangular passes run on four SIMDs). Beyond BIG_WG4_MAX_BATCH instances the one-wavefront kernel's four instances per CU win.
    // PMPC_BIG_WG4 = 0 / 1: developer switch (never / whenever eligible).
    unsigned threads = WAVE;
    if constexpr (Model::NG == 0 && Model::NP == 0) {
        const int wg4_on = pmpc_internal_switch(ctx, PMPC_SW_BIG_WG4_ON), wg4_off = pmpc_internal_switch(ctx, PMPC_SW_BIG_WG4_OFF);
        const bool eligible = Kws && lkern == sqp_kernel<Model, 0, 0, false, 0, true> && ss->kkt_form == 0 && ss->preconditioner == 0 && ss->qp_solver == 0 && ss->regularisation != 1 &&
                              dm.n <= BIG_COND_MAX_ROWS && dm.m <= BIG_COND_MAX_ROWS && slice_iters == 0;
        // Up to one instance per CU the team kernel may use the whole register file (512 per lane); from there to TWO instances per CU a second build of it,
        // compiled for 256 registers so that two workgroups share a CU, still beats one wavefront per instance (kite-sized, 512 instances: 11.7 against 13.4 ms;
        // the register diet costs it 4 % at 256 instances, hence two builds); beyond that the one-wavefront kernel's throughput wins (EXPERIMENTS.md round 5).
        const int per_cu = BIG_WG4_MAX_BATCH * (pmpc_internal_simd_count(ctx) / 4) / 256;
        const bool want = (wg4_on || wg4_off) ? (wg4_on != 0) : (B <= 2 * per_cu);
        if (eligible && want) {
            lkern = (B <= per_cu) ? sqp_kernel<Model, 0, 0, false, 0, true, false, false, false, true> : sqp_kernel<Model, 0, 0, false, 0, true, true, false, false, true>;
            threads = 4 * WAVE;
        }
    }
    const bool team_kernel = threads != WAVE;
    if constexpr (LDS_PATH_PROFILED<Model>::value) {   // developer builds with phase timers
        if (phase) {
            lkern = Kws ? sqp_kernel<Model, 0, 0, true, 0, true> : sqp_kernel<Model, 0, 0, true>;
            if constexpr (Model::NG == 0 && Model::NP == 0) { if (team_kernel) lkern = sqp_kernel<Model, 0, 0, true, 0, true, false, false, false, true>; }
        }
    } else if (phase && team_kernel) { lkern = sqp_kernel<Model, 0, 0, false, 0, true>; threads = WAVE; }
    if (hipFuncSetAttribute((const void*)lkern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PMPC_ERR_HIP;
    const int slice = (slice_iters > 0) ? slice_iters : ss->max_iter;
    for (int it = 0; it < ss->max_iter; it += slice)
        hipLaunchKernelGGL(lkern, dim3(B), dim3(threads), lds, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, *ss, *qs, Hws, Aws,
                           x, lam, info, phase, Kws, it, it + slice, slice_state, (unsigned)(lds / sizeof(double)));
    if (Kws && ss->kkt_form == 0 && ss->qp_solver == 0 && !pmpc_internal_switch(ctx, PMPC_SW_NO_REDO_LAUNCH)) {
        // redo launch (large-instance kernel, condensed mode): the instances whose QP gave up at its conditioning gate (PMPC_FLAG_ILLCOND; none on any
        // BASELINE workload) are solved again, from their guesses, by the same kernel in the (n + m)-row KKT form — every other workgroup reads one word and exits
        pmpc_sqp_settings ss_full = *ss; ss_full.kkt_form = 1;
        // (the two-wavefronts-per-SIMD build of the one-wavefront kernel: an instantiation of its own, so that a profile lists the redo launches — which
        //  normally do nothing — on their own line instead of halving the large-instance kernel's averages; the full KKT form has no team version)
        auto rkern = sqp_kernel<Model, 0, 0, false, 0, true, true>;
        if (hipFuncSetAttribute((const void*)rkern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PMPC_ERR_HIP;
        hipLaunchKernelGGL(rkern, dim3(B), dim3(WAVE), lds, stream, mdl, cd, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss_full, *qs, Hws, Aws,
                           x, lam, info, (unsigned long long*)nullptr, Kws, PMPC_REDO_MODE, ss->max_iter, (double*)nullptr, (unsigned)(lds / sizeof(double)));
    }
    return (hipGetLastError() == hipSuccess) ? PMPC_OK : PMPC_ERR_HIP;
}

}  // namespace pmpc
