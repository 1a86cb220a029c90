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

struct NoJView {};   // tag: the QP has no structure information (the plain QP entry points)

template <class Model, int NNODES>
struct JView {
    static_assert(NNODES > 0, "the node count is a compile-time constant of the register-resident kernels");
    enum { NX = Model::NX, NU = Model::NU, NP = Model::NP, NG = Model::NG, NDER = NX + NU + NP };
    static constexpr int VARX = NX * NNODES, VARU = NU * NNODES, ME = NX * NNODES, MI = NG * NNODES;
    const double* D;      // (P+1) x (P+1), column-major (LDS)
    const int* nsr;       // per node: packed (flags, segment, row), Ocp::stage_constants
    const double* jblk;   // [(k*NX + q)*NDER + i] = J(k*NX + q, gidx(k, i))
    const double* gblk;   // [(k*NG + g)*NDER + i] = J(ME + k*NG + g, gidx(k, i))
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
        const double* blk = jblk + (k * NX + q) * NDER;
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
            const double* gb = gblk + ri * NDER;
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
            for (int q = 0; q < NX; ++q) { bv[q] = jblk[(jn * NX + q) * NDER + dcol]; yv[q] = ys[jn * NX + q]; }
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
                for (int u = 0; u < 8; ++u) { const int r = (r0 + u < ME) ? r0 + u : 0; bv[u] = jblk[r * NDER + dcol]; yv[u] = ys[r]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (r0 + u < ME) b += bv[u] * yv[u];
            }
            a = isp ? b : a;
        }
        if constexpr (NG > 0) {   // inequality rows: node jn's (every node's for a parameter column)
            double g1 = a;
#pragma unroll
            for (int g = 0; g < NG; ++g) g1 += gblk[(jn * NG + g) * NDER + dcol] * ys[ME + jn * NG + g];
            if constexpr (NP > 0) {
                double g2 = a;
#pragma unroll
                for (int r = 0; r < MI; ++r) g2 += gblk[r * NDER + dcol] * ys[ME + r];
                a = isp ? g2 : g1;
            } else a = g1;
        }
        return a;
    }
};

}  // namespace pmpc
