// polympc_amd — block-sparse view of the collocation Jacobian, served from LDS (register-resident SQP kernels).
//
// The reference assembles J = [J_eq ; J_ineq] as a dense (NX*nn + NG*nn) x n matrix (continuous_ocp.hpp:797-878, :546-575), but only
//   * D(row_k, t) on the column (node kb+t, state q) of equality row (k, q), for the P+1 nodes of the segment that produces node k
//     (:817-827; last node: -reverse(first block row), :845-846),
//   * the own-node block  J((k,q), gidx(k,i)) = [D self entry on i == q] - t_scale * df_q/d(x,u,p)_i  (:870-872), and
//   * dg_g/d(x,u,p)_i on the inequality rows (k, g)  (:546-575)
// are structurally non-zero. The dense products the QP and the SQP form with J — A x and A' y in boxADMM's residuals
// (box_admm.hpp:398-415), J' lam in the Lagrangian gradient (continuous_ocp.hpp:2112-2114) — are sums over ascending column / row
// index of products  J(r,c) * v_c ; a structural zero contributes  +-0 , which leaves a finite partial sum unchanged bit for bit (the
// partial sums of these chains are never -0: they start at +0, and x + (-x) = +0 in round-to-nearest). rowdot / coldot below form
// exactly the non-zero products of those chains in the same ascending order from the per-node blocks (`jblk`, `gblk`: the values
// assemble_first_order stores into J, kept in LDS) and the differentiation matrix that is resident in LDS anyway — no HBM traffic.
// With a non-finite operand (0 * inf = NaN) the dense chain and the sparse one differ, so the callers test their operand first and
// keep the dense loops for that (flagged, PMPC_FLAG_NONFINITE) case.
//
// Both products walk the NNODES nodes with a compile-time trip count (fully unrolled: every LDS address is a per-lane base plus an
// immediate offset, all loads of a product are independent); a lane's own position decides with selects — no divergent branches —
// which of the candidate products enter its chain, and masked-out candidates enter as +0.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_ocp.hpp"

namespace pmpc {

// Row stride (in doubles) of an LDS table of 8-byte entries whose ROWS are read at per-lane row bases (the D~ tables of the condensed and the
// block-structured QP kernels: lane -> the row of its node): even, so that 16-byte reads stay aligned, and with stride / 2 ODD, so that 16
// consecutive rows start on 16 distinct 16-byte slots of the 256-byte bank row (ds_read_b64: distinct 8-byte positions over a 32-lane group;
// ds_read_b128: distinct slots over a 16-lane group). A power-of-two stride of 16 doubles put every row of the reference's 16-node grid on the
// same two bank pairs: 17.7 % of the wave cycles in SQ_LDS_BANK_CONFLICT (profiles/r04_cfgR128_pmc_summary.json). The arithmetic does not change.
__host__ __device__ constexpr int lds_row_stride(int len) { return (((len + (len & 1)) >> 1) & 1) ? len + (len & 1) : len + (len & 1) + 2; }


template <class Model, int NNODES>
struct JView {
    static_assert(NNODES > 0, "the node count is a compile-time constant of the register-resident kernels");
    enum { NX = Model::NX, NU = Model::NU, NP = Model::NP, NG = Model::NG, NDER = NX + NU + NP, JBS = OcpDims<Model>::JBS /* row stride of jblk / gblk (odd) */ };
    static constexpr int VARX = NX * NNODES, VARU = NU * NNODES, ME = NX * NNODES, MI = NG * NNODES;
    const double* D;      // (P+1) x (P+1), column-major (LDS)
    const int* nsr;       // per node: packed (flags, segment, row), Ocp::stage_constants
    const double* jblk;   // [(k*NX + q)*JBS + i] = J(k*NX + q, gidx(k, i))
    const double* gblk;   // [(k*NG + g)*JBS + i] = J(ME + k*NG + g, gidx(k, i))
    int P;
    // A scalar zero the optimiser cannot see through, added to P where a product starts: everything derived from P below (segment starts, D
    // offsets) is wave-uniform and loop-invariant, and would otherwise be computed once in the kernel prologue and live in — or be spilled
    // from — registers for the whole solve; reloading such values here costs an exposed memory round trip each (compiler hazard 1)
    __device__ __forceinline__ static int opaque_szero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }

    // sum_c J(r, c) xs[c], columns ascending (r in [0, m)); xs: n entries in LDS. Every lane passes a valid r.
    __device__ __forceinline__ double rowdot(int r, const double* xs) const {
        const bool eq = (NG == 0) || r < ME;
        const int re = eq ? r : 0;
        const int k = re / NX, q = re - k * NX;
        const int fl = nsr[k];
        const int seg = (fl >> 5) & 0xffffff, row = fl & 31;
        const bool last = k == NNODES - 1;
        const int P = this->P + opaque_szero();
        const int kb = seg * P, P1 = P + 1;
        const double* blk = jblk + (k * NX + q) * JBS;
        const double* xq = xs + q;
        // the row's D entries: D(row, t) at row + t (P+1); the last node's row -D(0, P - t) is stored behind D (OcpLds::D) at (P+1)^2 + t
        const double* drow = D + (last ? P1 * P1 : row);
        const int dstride = last ? 1 : P1;
        // candidates: one D entry per node j of the grid (live when j lies in the row's segment and is not the own node)
        double term[NNODES];
#pragma unroll
        for (int j = 0; j < NNODES; ++j) {
            const int t = j - kb;
            const bool inseg = (unsigned)t <= (unsigned)P;
            const int tc = inseg ? t : 0;
            const double dv = drow[tc * dstride];
            const double xv = xq[j * NX];
            const double pr = dv * xv;
            term[j] = inseg ? pr : 0.0;
        }
        double bv[NDER], xb[NDER];
#pragma unroll
        for (int i = 0; i < NX; ++i) { bv[i] = blk[i]; xb[i] = xs[k * NX + i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) { bv[NX + i] = blk[NX + i]; xb[NX + i] = xs[VARX + k * NU + i]; }
#pragma unroll
        for (int i = 0; i < NP; ++i) { bv[NX + NU + i] = blk[NX + NU + i]; xb[NX + NU + i] = xs[VARX + VARU + i]; }
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NNODES; ++j) a += (j < k) ? term[j] : 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) a += bv[i] * xb[i];
#pragma unroll
        for (int j = 0; j < NNODES; ++j) a += (j > k) ? term[j] : 0.0;
#pragma unroll
        for (int i = NX; i < NDER; ++i) a += bv[i] * xb[i];
        if constexpr (NG > 0) {
            const int ri = eq ? 0 : r - ME;
            const int kg = ri / NG;
            const double* gb = gblk + ri * JBS;
            double b = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) b += gb[i] * xs[kg * NX + i];
#pragma unroll
            for (int i = 0; i < NU; ++i) b += gb[NX + i] * xs[VARX + kg * NU + i];
#pragma unroll
            for (int i = 0; i < NP; ++i) b += gb[NX + NU + i] * xs[VARX + VARU + i];
            a = eq ? a : b;
        }
        return a;
    }

    // sum_r J(r, c) ys[r], rows ascending (c in [0, n)); ys: m entries in LDS. Every lane passes a valid c; `active` marks the lanes whose
    // result is used: when none of them holds a state column the D candidates are skipped (wave-uniform branch).
    __device__ __forceinline__ double coldot(int c, const double* ys, bool active = true) const {
        const bool xcol = c < VARX;
        const bool isp = (NP > 0) && c >= VARX + VARU;
        const int cu = c - VARX;
        const int jx = c / NX, ju = (!xcol && !isp) ? cu / NU : 0;
        const int jn = xcol ? jx : ju;                                   // node of the own block (parameter column: none)
        const int dcol = xcol ? c - jx * NX : (isp ? NX + NU + (cu - VARU) : NX + (cu - ju * NU));
        const int qx = xcol ? dcol : 0;
        const int P = this->P + opaque_szero();
        const int P1 = P + 1;
        double a = 0.0;
        double term[NNODES];
        const bool any_x = __builtin_amdgcn_ballot_w64(active && xcol) != 0;   // wave-uniform
        if (any_x) {
            const double* yq = ys + qx;
            // (segment start kb_k, D row row_k) of node k — Ocp::seg_row — tracked with wave-uniform scalar arithmetic instead of read from the
            // node table: the D loads of all nodes then depend on nothing but the lane's own column and are issued together
            int kb = 0, row = 0;
#pragma unroll
            for (int k = 0; k < NNODES; ++k) {   // equality row (k, qx): D(row_k, jn - kb_k) when node jn lies in the segment that produces node k
                const int kbk = (k == NNODES - 1) ? (NNODES - 1) - P : kb;
                const int t = jn - kbk;
                const bool inseg = xcol && (unsigned)t <= (unsigned)P;
                const int tc = inseg ? t : 0;
                const int di = (k == NNODES - 1) ? P1 * P1 + tc : row + tc * P1;
                const double dv = D[di];
                const double yv = yq[k * NX];
                const double pr = dv * yv;
                term[k] = inseg ? pr : 0.0;
                ++row;
                if (row == P) { row = 0; kb += P; }
            }
#pragma unroll
            for (int k = 0; k < NNODES; ++k) a += (k < jn) ? term[k] : 0.0;
        }
        {   // own-node block: the NX equality rows of node jn
            double bv[NX > 0 ? NX : 1], yv[NX > 0 ? NX : 1];
#pragma unroll
            for (int q = 0; q < NX; ++q) { bv[q] = jblk[(jn * NX + q) * JBS + dcol]; yv[q] = ys[jn * NX + q]; }
#pragma unroll
            for (int q = 0; q < NX; ++q) { const double pr = bv[q] * yv[q]; a += isp ? 0.0 : pr; }
        }
        if (any_x) {
#pragma unroll
            for (int k = 0; k < NNODES; ++k) a += (k > jn) ? term[k] : 0.0;
        }
        if constexpr (NP > 0) {   // parameter column: dense over the equality rows
            double b = 0.0;
#pragma unroll
            for (int r0 = 0; r0 < ME; r0 += 8) {
                double bv[8], yv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int r = (r0 + u < ME) ? r0 + u : 0; bv[u] = jblk[r * JBS + dcol]; yv[u] = ys[r]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (r0 + u < ME) b += bv[u] * yv[u];
            }
            a = isp ? b : a;
        }
        if constexpr (NG > 0) {   // inequality rows: node jn's (every node's for a parameter column)
            double g1 = a;
#pragma unroll
            for (int g = 0; g < NG; ++g) g1 += gblk[(jn * NG + g) * JBS + dcol] * ys[ME + jn * NG + g];
            if constexpr (NP > 0) {
                double g2 = a;
#pragma unroll
                for (int r = 0; r < MI; ++r) g2 += gblk[r * JBS + dcol] * ys[ME + r];
                a = isp ? g2 : g1;
            } else a = g1;
        }
        return a;
    }
};

// The same view for a run-time node count and any memory (the large-instance kernel keeps the blocks and the collocation constants in its HBM
// scratch): the entry J(r, c) itself — the operands of the matrix-core product A' diag(rho) A — and the two sparse products of the condensed
// linear solve (pmpc_qp_big.hpp), which are fma chains over the entries that are not exactly zero, rows / columns ascending: the restatement
// (the CPU restatement of the test suite, policy PIVOT_CONDENSED) walks the dense matrix and skips its zeros, so structural zeros and numerical zeros are the same statement on
// both sides, whatever the operand.
template <class Model>
struct JViewRT {
    enum { NX = Model::NX, NU = Model::NU, NP = Model::NP, NG = Model::NG, NDER = NX + NU + NP, JBS = OcpDims<Model>::JBS /* row stride of jblk / gblk (odd) */ };
    const double* D;      // (P+1) x (P+1) column-major, followed by the last node's row (OcpLds::D)
    const int* nsr;
    const double* jblk;
    const double* gblk;
    int P, NNo, VARX, VARU, ME;
    const double* tab = nullptr;   // (NG = NP = 0) four dense (NNo + 1) x NNP tables of D~ in LDS, see build_tables; null: the products below walk the structure

    struct Node { int kb; const double* drow; int dstride; };
    // segment start and D row of the equality rows of node k (Ocp::seg_row) from arithmetic alone — k / P through a float reciprocal, exact for the
    // node counts in question — so that the D loads below depend on no other load (the node table sits in HBM here: a dependent round trip each)
    __device__ __forceinline__ Node node(int k) const {
        const bool last = k == NNo - 1;
        const int seg = (int)(((float)k + 0.5f) * (1.0f / (float)P));
        const int row = k - seg * P;
        Node nd; nd.kb = last ? (NNo - 1) - P : seg * P; nd.drow = D + (last ? (P + 1) * (P + 1) : row); nd.dstride = last ? 1 : P + 1;
        return nd;
    }
    struct Col { bool xcol, pcol; int jn, dcol; };
    __device__ __forceinline__ Col column(int c) const {
        Col k; k.xcol = c < VARX; k.pcol = (NP > 0) && c >= VARX + VARU;
        const int cu = c - VARX;
        const int jx = c / NX, ju = (!k.xcol && !k.pcol) ? cu / NU : 0;
        k.jn = k.xcol ? jx : ju;
        k.dcol = k.xcol ? c - jx * NX : (k.pcol ? NX + NU + (cu - VARU) : NX + (cu - ju * NU));
        return k;
    }
    struct Row { bool eq; int k, q, ri, kg; Node nd; };
    __device__ __forceinline__ Row rowinfo(int r) const {
        Row w; w.eq = (NG == 0) || r < ME;
        const int re = w.eq ? r : 0;
        w.k = re / NX; w.q = re - w.k * NX; w.nd = node(w.k);
        w.ri = w.eq ? 0 : r - ME; w.kg = (NG > 0) ? w.ri / (NG > 0 ? NG : 1) : 0;
        return w;
    }
    // J(r, c) for a decoded row and column
    __device__ __forceinline__ double jval(const Row& w, const Col& cc) const {
        const int t = cc.jn - w.nd.kb;
        const bool inseg = cc.xcol && cc.dcol == w.q && (unsigned)t <= (unsigned)P;
        const double dv = w.nd.drow[(inseg ? t : 0) * w.nd.dstride];
        const double bv = jblk[(w.k * NX + w.q) * JBS + cc.dcol];
        double v = (cc.pcol || cc.jn == w.k) ? bv : (inseg ? dv : 0.0);
        if constexpr (NG > 0) {
            const double gv = gblk[w.ri * JBS + cc.dcol];
            const double vg = (cc.pcol || cc.jn == w.kg) ? gv : 0.0;
            v = w.eq ? v : vg;
        }
        return v;
    }
    __device__ __forceinline__ double jval(int r, int c) const { return jval(rowinfo(r), column(c)); }
    // is J(r, c) a structural non-zero? (no memory access)
    __device__ __forceinline__ bool structural(const Row& w, const Col& cc) const {
        const int t = cc.jn - w.nd.kb;
        const bool inseg = cc.xcol && cc.dcol == w.q && (unsigned)t <= (unsigned)P;
        const int kk = w.eq ? w.k : w.kg;
        return cc.pcol || cc.jn == kk || (w.eq && inseg);
    }
    __device__ __forceinline__ static double step(double a, double v, double x, bool on) { const double f = fma(v, x, a); return (on && v != 0.0) ? f : a; }

    // ---- table-driven form of the two products of the condensed solve (round 4). D~(r, k): the differentiation-matrix entry of equality row node r on the
    // state columns of node k, 0 on the own node and outside the row's segment. Four tables, rows padded to a multiple of 8 entries, one all-zero row behind
    // each (control columns): [0] column node jn, row nodes below it (k < jn)  [1] column node jn, k > jn  [2] row node k, column nodes j < k  [3] j > k.
    // The products then are fma chains with coefficients at per-lane bases + immediate offsets — the entries before the own node's block, the block, the
    // entries behind it: the ascending order of the structure-walking versions (a zero coefficient leaves a finite partial sum unchanged) without their
    // index arithmetic and selects (43 k + 20 k cycles per ADMM iteration of config C for the two products; with the tables: 4 k + 3 k).
    __host__ __device__ static int tab_len(int nno) { return (nno + 7) & ~7; }                 // entries read per row (batches of 8; the padding is zeros)
    __host__ __device__ static int tab_nnp(int nno) { return lds_row_stride(tab_len(nno)); }   // row stride (bank-conflict-free per-lane row bases)
    __host__ __device__ static size_t tab_doubles(int nno) { return (size_t)4 * (nno + 1) * tab_nnp(nno); }
    // the tables grow with the square of the node count: up to 16 nodes (8.5 KB) they are worth their LDS; beyond, the occupancy of the two-wavefront
    // builds pays for them (21 nodes: 17 KB, five instead of eight instances per CU) and the products walk the structure as before
    __host__ __device__ static bool tab_worth_it(int nno) { return nno <= 16; }
    __device__ __forceinline__ static void build_tables(const double* Dm, int P, int NNo, double* tb) {
        const int NNP = tab_nnp(NNo), TS = (NNo + 1) * NNP, P1 = P + 1;
        for (int e = lane_id(); e < 4 * TS; e += WAVE) tb[e] = 0.0;
        wsync();
        for (int e = lane_id(); e < NNo * NNo; e += WAVE) {
            const int r = e / NNo, k = e - r * NNo;
            const bool lastr = r == NNo - 1;
            const int kbr = lastr ? NNo - 1 - P : (r / P) * P;
            const int rowr = lastr ? P : r - kbr;
            const int t = k - kbr;
            const bool cpl = k != r && (unsigned)t <= (unsigned)P;
            const double dv = Dm[lastr ? P1 * P1 + (cpl ? t : 0) : rowr + (cpl ? t : 0) * P1];
            const double v = cpl ? dv : 0.0;
            tb[(r < k ? 0 : TS) + k * NNP + r] = v;               // column node k, row node r
            tb[(k < r ? 2 * TS : 3 * TS) + r * NNP + k] = v;      // row node r, column node k
        }
        wsync();
    }
    // init, then the fma chain over column c against us (rows ascending); bv: the own node's block column (col_block)
    __device__ __forceinline__ double coldot_fma_tab(const Col& cc, const double (&bv)[NX + NG > 0 ? NX + NG : 1], const double* us, double init) const {
        const int NNP = tab_nnp(NNo), NNL = tab_len(NNo), TS = (NNo + 1) * NNP;
        const double* lo = tab + (cc.xcol ? cc.jn : NNo) * NNP;
        const double* hi = lo + TS;
        const double* uq = us + (cc.xcol ? cc.dcol : 0);
        double a = init;
        for (int k0 = 0; k0 < NNL; k0 += 8) {
            double dv[8], uv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dv[u] = lo[k0 + u]; uv[u] = uq[((k0 + u < NNo) ? k0 + u : 0) * NX]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fma(dv[u], uv[u], a);
        }
        {
            double vv[NX];
#pragma unroll
            for (int q = 0; q < NX; ++q) vv[q] = us[cc.jn * NX + q];
#pragma unroll
            for (int q = 0; q < NX; ++q) a = fma(bv[q], vv[q], a);
        }
        for (int k0 = 0; k0 < NNL; k0 += 8) {
            double dv[8], uv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dv[u] = hi[k0 + u]; uv[u] = uq[((k0 + u < NNo) ? k0 + u : 0) * NX]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fma(dv[u], uv[u], a);
        }
        return a;
    }
    // fma chain over equality row r = (k, q) against xs (columns ascending), starting from 0; bv: the row's own-node block (row_block)
    __device__ __forceinline__ double rowdot_fma_tab(const Row& w, const double (&bv)[NDER], const double* xs) const {
        const int NNP = tab_nnp(NNo), NNL = tab_len(NNo), TS = (NNo + 1) * NNP;
        const double* lo = tab + 2 * TS + w.k * NNP;
        const double* hi = lo + TS;
        const double* xq = xs + w.q;
        double a = 0.0;
        for (int j0 = 0; j0 < NNL; j0 += 8) {
            double dv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dv[u] = lo[j0 + u]; xv[u] = xq[((j0 + u < NNo) ? j0 + u : 0) * NX]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fma(dv[u], xv[u], a);
        }
        double xb[NDER];
#pragma unroll
        for (int i = 0; i < NX; ++i) xb[i] = xs[w.k * NX + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) xb[NX + i] = xs[VARX + w.k * NU + i];
#pragma unroll
        for (int i = 0; i < NX; ++i) a = fma(bv[i], xb[i], a);
        for (int j0 = 0; j0 < NNL; j0 += 8) {
            double dv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dv[u] = hi[j0 + u]; xv[u] = xq[((j0 + u < NNo) ? j0 + u : 0) * NX]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fma(dv[u], xv[u], a);
        }
#pragma unroll
        for (int i = NX; i < NX + NU; ++i) a = fma(bv[i], xb[i], a);
        return a;
    }

    // ---- the two sparse products of the condensed solve. D (`D`) and the vectors live in LDS; the per-node blocks come from the HBM scratch and are
    // requested first (row_block / col_block, every pass of a product before any of them is consumed), so that a product costs one memory round trip.
    __device__ __forceinline__ void row_block(const Row& w, double (&bv)[NDER]) const {
        const double* blk = w.eq ? jblk + (w.k * NX + w.q) * JBS : gblk + w.ri * JBS;
#pragma unroll
        for (int i = 0; i < NDER; ++i) bv[i] = blk[i];
    }
    // fma chain over the non-zero entries of row r against xs (n entries), columns ascending, starting from 0
    __device__ __forceinline__ double rowdot_fma(const Row& w, const double (&bv)[NDER], const double* xs) const {
        const int kk = w.eq ? w.k : w.kg;
        double xb[NDER];
#pragma unroll
        for (int i = 0; i < NX; ++i) xb[i] = xs[kk * NX + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) xb[NX + i] = xs[VARX + kk * NU + i];
#pragma unroll
        for (int i = 0; i < NP; ++i) xb[NX + NU + i] = xs[VARX + VARU + i];
        double a = 0.0;
        for (int ph = 0; ph < 2; ++ph) {   // segment nodes before the own node, the own node's state columns, segment nodes after it
            for (int t0 = 0; t0 <= P; t0 += 8) {
                double dv[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int t = (t0 + u <= P) ? t0 + u : 0; dv[u] = w.nd.drow[t * w.nd.dstride]; xv[u] = xs[(w.nd.kb + t) * NX + w.q]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = w.nd.kb + t0 + u; a = step(a, dv[u], xv[u], w.eq && t0 + u <= P && (ph == 0 ? j < w.k : j > w.k)); }
            }
            if (ph == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) a = step(a, bv[i], xb[i], true);
            }
        }
#pragma unroll
        for (int i = NX; i < NDER; ++i) a = step(a, bv[i], xb[i], true);
        return a;
    }
    // ---- the same two products as multiply-add chains (a += v x over the STRUCTURAL entries, ascending): boxADMM's residual evaluation. The dense
    // chains they replace also add the structural zeros' +-0, which leaves a finite partial sum unchanged: callers test their operand for
    // non-finite values first and keep the dense loops for that case.
    __device__ __forceinline__ static double stepma(double a, double v, double x, bool on) { const double f = a + v * x; return on ? f : a; }
    __device__ __forceinline__ double rowdot_ma(const Row& w, const double (&bv)[NDER], const double* xs) const {
        const int kk = w.eq ? w.k : w.kg;
        double xb[NDER];
#pragma unroll
        for (int i = 0; i < NX; ++i) xb[i] = xs[kk * NX + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) xb[NX + i] = xs[VARX + kk * NU + i];
#pragma unroll
        for (int i = 0; i < NP; ++i) xb[NX + NU + i] = xs[VARX + VARU + i];
        double a = 0.0;
        for (int ph = 0; ph < 2; ++ph) {
            for (int t0 = 0; t0 <= P; t0 += 8) {
                double dv[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int t = (t0 + u <= P) ? t0 + u : 0; dv[u] = w.nd.drow[t * w.nd.dstride]; xv[u] = xs[(w.nd.kb + t) * NX + w.q]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = w.nd.kb + t0 + u; a = stepma(a, dv[u], xv[u], w.eq && t0 + u <= P && (ph == 0 ? j < w.k : j > w.k)); }
            }
            if (ph == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) a = stepma(a, bv[i], xb[i], true);
            }
        }
#pragma unroll
        for (int i = NX; i < NDER; ++i) a = stepma(a, bv[i], xb[i], true);
        return a;
    }
    static constexpr int NCB = NX + NG;   // entries of a column inside its own node's rows (equality rows, then inequality rows)
    __device__ __forceinline__ void col_block(const Col& cc, double (&bv)[NCB > 0 ? NCB : 1]) const {
#pragma unroll
        for (int q = 0; q < NX; ++q) bv[q] = jblk[(cc.jn * NX + q) * JBS + cc.dcol];
#pragma unroll
        for (int g = 0; g < NG; ++g) bv[NX + g] = gblk[(cc.jn * NG + g) * JBS + cc.dcol];
    }
    // init, then the fma chain over the non-zero entries of column c against us (m entries), rows ascending
    __device__ __forceinline__ double coldot_fma(const Col& cc, const double (&bv)[NCB > 0 ? NCB : 1], const double* us, double init) const {
        const int qx = cc.xcol ? cc.dcol : 0;
        double a = init;
        for (int ph = 0; ph < 2; ++ph) {
            for (int k0 = 0; k0 < NNo; k0 += 8) {   // equality rows (k, qx) whose segment holds node jn, above / below the own node
                double dv[8], uv[8]; bool on[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = (k0 + u < NNo) ? k0 + u : 0;
                    const Node nd = node(k);
                    const int t = cc.jn - nd.kb;
                    const bool inseg = cc.xcol && (unsigned)t <= (unsigned)P;
                    dv[u] = nd.drow[(inseg ? t : 0) * nd.dstride];
                    uv[u] = us[k * NX + qx];
                    on[u] = inseg && (k0 + u < NNo) && (ph == 0 ? k < cc.jn : k > cc.jn);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) a = step(a, dv[u], uv[u], on[u]);
            }
            if (ph == 0) {   // own-node block: the NX equality rows of node jn (parameter column: every equality row, below)
#pragma unroll
                for (int q = 0; q < NX; ++q) a = step(a, bv[q], us[cc.jn * NX + q], !cc.pcol);
            }
        }
        if constexpr (NP > 0) {
            double b = init;
            for (int r0 = 0; r0 < ME; r0 += 8) {
                double pv[8], uv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int r = (r0 + u < ME) ? r0 + u : 0; pv[u] = jblk[r * JBS + cc.dcol]; uv[u] = us[r]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) b = step(b, pv[u], uv[u], r0 + u < ME);
            }
            a = cc.pcol ? b : a;
        }
        if constexpr (NG > 0) {
            double g1 = a;
#pragma unroll
            for (int g = 0; g < NG; ++g) g1 = step(g1, bv[NX + g], us[ME + cc.jn * NG + g], true);
            if constexpr (NP > 0) {
                double g2 = a;
                for (int r = 0; r < NG * NNo; ++r) g2 = step(g2, gblk[r * JBS + cc.dcol], us[ME + r], true);
                a = cc.pcol ? g2 : g1;
            } else a = g1;
        }
        return a;
    }
    __device__ __forceinline__ double coldot_ma(const Col& cc, const double (&bv)[NCB > 0 ? NCB : 1], const double* us) const {
        const int qx = cc.xcol ? cc.dcol : 0;
        double a = 0.0;
        for (int ph = 0; ph < 2; ++ph) {
            for (int k0 = 0; k0 < NNo; k0 += 8) {
                double dv[8], uv[8]; bool on[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = (k0 + u < NNo) ? k0 + u : 0;
                    const Node nd = node(k);
                    const int t = cc.jn - nd.kb;
                    const bool inseg = cc.xcol && (unsigned)t <= (unsigned)P;
                    dv[u] = nd.drow[(inseg ? t : 0) * nd.dstride];
                    uv[u] = us[k * NX + qx];
                    on[u] = inseg && (k0 + u < NNo) && (ph == 0 ? k < cc.jn : k > cc.jn);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) a = stepma(a, dv[u], uv[u], on[u]);
            }
            if (ph == 0) {
#pragma unroll
                for (int q = 0; q < NX; ++q) a = stepma(a, bv[q], us[cc.jn * NX + q], !cc.pcol);
            }
        }
        if constexpr (NP > 0) {
            double b = 0.0;
            for (int r0 = 0; r0 < ME; r0 += 8) {
                double pv[8], uv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int r = (r0 + u < ME) ? r0 + u : 0; pv[u] = jblk[r * JBS + cc.dcol]; uv[u] = us[r]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) b = stepma(b, pv[u], uv[u], r0 + u < ME);
            }
            a = cc.pcol ? b : a;
        }
        if constexpr (NG > 0) {
            double g1 = a;
#pragma unroll
            for (int g = 0; g < NG; ++g) g1 = stepma(g1, bv[NX + g], us[ME + cc.jn * NG + g], true);
            if constexpr (NP > 0) {
                double g2 = a;
                for (int r = 0; r < NG * NNo; ++r) g2 = stepma(g2, gblk[r * JBS + cc.dcol], us[ME + r], true);
                a = cc.pcol ? g2 : g1;
            } else a = g1;
        }
        return a;
    }
};

}  // namespace pmpc
