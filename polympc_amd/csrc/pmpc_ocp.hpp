// polympc_amd — Chebyshev-collocation transcription on the device, one wavefront per OCP instance.
//
// Replaces the DENSE members of ContinuousOCP (/root/reference/src/control/continuous_ocp.hpp): equalities :739-766,
// inequalities :770-782, equalities_linearised :797-878, _inequalities_linearised_dense :546-575, cost :1182-1207,
// cost_gradient :1210-1249, cost_gradient_hessian :1256-1367, lagrangian_gradient :1960-1975,
// lagrangian_gradient_hessian :2100-2174.
//
// Mapping: lane k evaluates collocation node k (dynamics / Lagrange term / path constraints with forward AD) and
// stages the per-node values and derivative blocks in LDS; the wave then assembles the dense Jacobian (into the
// instance's HBM workspace, column-major m x n) and Hessian (n x n). The Chebyshev differentiation matrix, the
// Clenshaw–Curtis weights and the time grid are staged once per workgroup in LDS. Second derivatives are taken one
// seed direction at a time (Dual<Dual<double,NDER>,1>) to keep the register footprint at 2*(NDER+1) doubles per AD
// variable; every Hessian entry goes through the same operation sequence as the reference's nested AutoDiffScalar.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_ad.hpp"
#include "pmpc_models.hpp"
#include "pmpc_qp.hpp"

namespace pmpc {

constexpr int MAX_P = 15;      // polynomial order per segment
constexpr int MAX_NODES = 64;  // P*S+1

// host-computed collocation constants (pmpc_chebyshev / pmpc_cheb.hpp), resident in HBM, staged to LDS per workgroup
struct ChebData {
    int P, S, NN, _pad;
    double t_start, t_stop, t_scale;
    double D[(MAX_P + 1) * (MAX_P + 1)];  // column-major (P+1)x(P+1)
    double w[MAX_P + 1];
    double tn[MAX_NODES];                 // time nodes, descending (node 0 = t_stop)
};

template <class Model>
struct OcpDims {
    enum { NX = Model::NX, NU = Model::NU, NP = Model::NP, ND = Model::ND, NG = Model::NG, NDER = NX + NU + NP,
           JBS = NDER | 1 };   // row stride of the per-node blocks of J kept beside the dense matrix (jblk / gblk, pmpc_jview.hpp): ODD, so that the rows of 32
                               // consecutive lanes start on 32 distinct 8-byte bank positions (an even NDER put config B's rows on 16: 6 % of its wave cycles in bank conflicts)
    int NN, VARX, VARU, n, me, mi, m;
    __host__ __device__ OcpDims(int P, int S) {
        NN = P * S + 1; VARX = NX * NN; VARU = NU * NN; n = VARX + VARU + NP; me = VARX; mi = NG * NN; m = me + mi;
    }
    __host__ __device__ int gidx(int k, int i) const {
        return i < NX ? k * NX + i : (i < NX + NU ? VARX + k * NU + (i - NX) : VARX + VARU + (i - NX - NU));
    }
};

// per-instance LDS staging for the transcription
template <class Model>
struct OcpLds {
    using Dm = OcpDims<Model>;
    enum { NX = Dm::NX, NU = Dm::NU, NP = Dm::NP, NG = Dm::NG, NDER = Dm::NDER };
    double *D, *w, *tn;                                 // collocation constants; D is followed by Dl[t] = -D(0, P - t), the last node's row of J (:845-846)
    double *nd, *nw1, *nw2; int* nsr;                   // per node: D self entry, the two quadrature weights * t_scale, packed (flags, segment, row)
    double *fval, *fjac, *Lval, *Lgrad, *gval, *gjac;   // per node: f (NX), df (NX*NDER), L, dL (NDER), g (NG), dg (NG*NDER)
    double *Lhes, *dhes;                                // per node: d2L (NDER^2), sum lam * d2(f,g) (NDER^2)
    double *Mval, *Mgrad, *Mhes;                        // Mayer term at node 0
    double *DX;                                         // D*X per node (NX)
    __host__ __device__ static size_t doubles(int P, int S) {
        const int NN = P * S + 1;
        return const_doubles(P, S) + (size_t)NN * (NX + NX * NDER + 1 + NDER + NG + NG * NDER + 2 * NDER * NDER + NX) +
               1 + NDER + NDER * NDER + 8;
    }
    // collocation constants come first; everything after them is per-linearisation staging
    __host__ __device__ static size_t const_doubles(int P, int S) {
        const int NN = P * S + 1;
        return (size_t)(P + 1) * (P + 1) + (P + 1) /*Dl*/ + (P + 1) + NN + 3 * (size_t)NN + (size_t)(NN + 1) / 2;
    }
    __device__ __forceinline__ double* carve(double* p, int P, int S) {
        const int NN = P * S + 1;
        D = p; p += (P + 1) * (P + 1) + (P + 1); w = p; p += P + 1; tn = p; p += NN;
        nd = p; p += NN; nw1 = p; p += NN; nw2 = p; p += NN; nsr = (int*)p; p += (NN + 1) / 2;
        fval = p; p += NN * NX; fjac = p; p += NN * NX * NDER; Lval = p; p += NN; Lgrad = p; p += NN * NDER;
        gval = p; p += NN * NG; gjac = p; p += NN * NG * NDER; Lhes = p; p += NN * NDER * NDER; dhes = p; p += NN * NDER * NDER;
        Mval = p; p += 1; Mgrad = p; p += NDER; Mhes = p; p += NDER * NDER; DX = p; p += NN * NX;
        return p;
    }
};

template <class Model>
struct Ocp {
    using Dm = OcpDims<Model>;
    enum { NX = Dm::NX, NU = Dm::NU, NP = Dm::NP, ND = Dm::ND, NG = Dm::NG, NDER = Dm::NDER };
    using ad1 = Dual<double, NDER>;
    using ad2c = Dual<ad1, 1>;  // one outer seed direction at a time

    const Model& model;
    Dm dm;
    int P, S;
    double ts;
    OcpLds<Model> s;
    const double* d;  // static parameters of this instance (ND)
    // block-sparse copy of J for pmpc_jview.hpp (register-resident kernels): the per-node blocks assemble_first_order writes into J, kept in
    // an LDS region that outlives the per-node staging (which the QP's staging aliases); keep_blk says whether the two pointers are set
    double* jblk = nullptr; double* gblk = nullptr; bool keep_blk = false;
    double* jtab = nullptr;   // large-instance mode: the D~ tables of the condensed solve's sparse products (JViewRT::build_tables), LDS
    double* Dlds = nullptr;   // large-instance mode: an LDS copy of D (+ the last node's row) for the condensed linear algebra (the constants themselves sit in the HBM scratch there)

    __device__ Ocp(const Model& mdl, int P_, int S_, double t_scale) : model(mdl), dm(P_, S_), P(P_), S(S_), ts(t_scale), d(nullptr) {}

    __device__ __forceinline__ void stage_constants(const ChebData* cd) {
        const int ln = lane_id();
        for (int i = ln; i < (P + 1) * (P + 1); i += WAVE) s.D[i] = cd->D[i];
        for (int i = ln; i <= P; i += WAVE) s.D[(P + 1) * (P + 1) + i] = -cd->D[(P - i) * (P + 1)];
        for (int i = ln; i <= P; i += WAVE) s.w[i] = cd->w[i];
        for (int i = ln; i < dm.NN; i += WAVE) s.tn[i] = cd->tn[i];
        // per-node table: everything the hot loops would otherwise derive from k / P and k % P (integer divisions by a run-time
        // value) — (segment, row) of seg_row, the node's own D entry, and the cost-gradient weights of :1218-1240 with the
        // conditions under which they apply (bit 30: node closes a segment, bit 29: node opens / lies inside one)
        for (int k = ln; k < dm.NN; k += WAVE) {
            int seg, row;
            if (k == dm.NN - 1) { seg = S - 1; row = P; } else { seg = k / P; row = k % P; }
            const bool closes = (k % P == 0) && k > 0, inside = k < dm.NN - 1;
            s.nsr[k] = (closes ? (1 << 30) : 0) | (inside ? (1 << 29) : 0) | (seg << 5) | row;
            s.nd[k] = inside ? cd->D[row + row * (P + 1)] * 1.0 : -cd->D[0];
            s.nw1[k] = ts * cd->w[P];
            s.nw2[k] = ts * cd->w[k % P];
        }
        wsync();
    }

    // (segment, row) whose D row produces node k: later segments overwrite the junction row (:750-751)
    __device__ __forceinline__ void seg_row(int k, int& seg, int& row) const {
        const int v = s.nsr[k];
        seg = (v >> 5) & 0xffffff; row = v & 31;
    }

    // ---- values only: c = D*X - t_scale*f, g (equalities :739-766, inequalities :770-782)
    __device__ __forceinline__ void constraints(const double* var, double* c) {
        for (int k = lane_id(); k < dm.NN; k += WAVE) {
            double f[NX > 0 ? NX : 1];
            for (int q = 0; q < NX; ++q) f[q] = 0.0;
            const Value tk(s.tn[k]);
            model.template dynamics_impl<Value>(as_cvalues(var + k * NX), as_cvalues(var + dm.VARX + k * NU),
                                                as_cvalues(var + dm.VARX + dm.VARU), cref<double>(d), tk, as_values(f));
            int seg, row; seg_row(k, seg, row);
            for (int q = 0; q < NX; ++q) {
                double acc = 0.0;
                for (int j = 0; j <= P; ++j) acc += s.D[row + j * (P + 1)] * var[(seg * P + j) * NX + q];
                double cv = acc;
                cv -= ts * f[q];
                c[k * NX + q] = cv;
            }
            if (NG > 0) {
                double g[NG > 0 ? NG : 1];
                for (int q = 0; q < NG; ++q) g[q] = 0.0;
                model.template inequality_constraints_impl<Value>(as_cvalues(var + k * NX), as_cvalues(var + dm.VARX + k * NU),
                                                                  as_cvalues(var + dm.VARX + dm.VARU), cref<double>(d), tk.v, as_values(g));
                for (int q = 0; q < NG; ++q) c[dm.me + k * NG + q] = g[q];
            }
        }
        wsync();
    }

    // ---- cost (:1182-1207)
    __device__ __forceinline__ double cost(const double* var) {
        for (int k = lane_id(); k < dm.NN; k += WAVE) {
            Value L(0.0);
            model.template lagrange_term_impl<Value>(as_cvalues(var + k * NX), as_cvalues(var + dm.VARX + k * NU),
                                                     as_cvalues(var + dm.VARX + dm.VARU), cref<double>(d), s.tn[k], L);
            s.Lval[k] = L.v;
        }
        wsync();
        double c = 0.0;
        for (int sg = 0; sg < S; ++sg)
            for (int k = 0; k <= P; ++k) c += ts * s.w[k] * s.Lval[sg * P + k];
        Value M(0.0);
        model.template mayer_term_impl<Value>(as_cvalues(var), as_cvalues(var + dm.VARX), as_cvalues(var + dm.VARX + dm.VARU),
                                              cref<double>(d), s.tn[0], M);
        c += M.v;
        wsync();
        return c;
    }

    template <class T> __device__ void seed1(const double* var, int k, T* x, T* u, T* p) const {
        int idx = 0;
        for (int i = 0; i < NX; ++i, ++idx) { x[i] = T(var[k * NX + i]); x[i].d[idx] = 1.0; }
        for (int i = 0; i < NU; ++i, ++idx) { u[i] = T(var[dm.VARX + k * NU + i]); u[i].d[idx] = 1.0; }
        for (int i = 0; i < NP; ++i, ++idx) { p[i] = T(var[dm.VARX + dm.VARU + i]); p[i].d[idx] = 1.0; }
    }
    // second-order seeding for outer direction `dir` (continuous_ocp.hpp:691-735 restricted to one outer partial)
    __device__ void seed2(const double* var, int k, int dir, ad2c* x, ad2c* u, ad2c* p) const {
        int idx = 0;
        auto mk = [&](double val, int id) { ad2c r; r.v = ad1(val); r.v.d[id] = 1.0; r.d[0] = ad1(id == dir ? 1.0 : 0.0); return r; };
        for (int i = 0; i < NX; ++i, ++idx) x[i] = mk(var[k * NX + i], idx);
        for (int i = 0; i < NU; ++i, ++idx) u[i] = mk(var[dm.VARX + k * NU + i], idx);
        for (int i = 0; i < NP; ++i, ++idx) p[i] = mk(var[dm.VARX + dm.VARU + i], idx);
    }

    // ---- per-node first-order stage: f, df, L, dL, g, dg, DX, Mayer value+gradient
    // One lane per (node, derivative direction): the NDER partial derivatives of a node are independent columns of the same
    // forward sweep, so each lane carries ONE derivative component (Dual<double,1>: 4 operations per product instead of
    // 1 + 3*NDER) and NN*NDER lanes work side by side. Every value and every derivative goes through exactly the operations
    // it went through as component `dir` of a Dual<double,NDER>; the lanes of direction 0 store the values.
    __device__ __forceinline__ void stage_first_order(const double* var) { stage_first_order_part<1>(var, 0); wsync(); }
    // the (node, direction) pairs kd = w * 64 + lane, stride 64 NW: wavefront w of a team of NW (BigTeam, pmpc_qp_big.hpp); the caller synchronises behind it
    template <int NW>
    __device__ __forceinline__ void stage_first_order_part(const double* var, int w) {
        using ad = Dual<double, 1>;
        for (int kd = lane_id() + WAVE * w; kd < dm.NN * NDER; kd += WAVE * NW) {
            const int k = kd % dm.NN, dir = kd / dm.NN;
            ad x[NX > 0 ? NX : 1], u[NU > 0 ? NU : 1], p[NP > 0 ? NP : 1], y[NX > 0 ? NX : 1];
            {
                int idx = 0;
                for (int i = 0; i < NX; ++i, ++idx) { x[i] = ad(var[k * NX + i]); x[i].d[0] = (idx == dir) ? 1.0 : 0.0; }
                for (int i = 0; i < NU; ++i, ++idx) { u[i] = ad(var[dm.VARX + k * NU + i]); u[i].d[0] = (idx == dir) ? 1.0 : 0.0; }
                for (int i = 0; i < NP; ++i, ++idx) { p[i] = ad(var[dm.VARX + dm.VARU + i]); p[i].d[0] = (idx == dir) ? 1.0 : 0.0; }
            }
            for (int q = 0; q < NX; ++q) y[q] = ad(0.0);
            ad tk(s.tn[k]);
            model.template dynamics_impl<ad>(cref<ad>(x), cref<ad>(u), cref<ad>(p), cref<double>(d), tk, vref<ad>(y));
            for (int q = 0; q < NX; ++q) {
                if (dir == 0) s.fval[k * NX + q] = y[q].v;
                s.fjac[(k * NX + q) * NDER + dir] = y[q].d[0];
            }
            ad L(0.0);
            model.template lagrange_term_impl<ad>(cref<ad>(x), cref<ad>(u), cref<ad>(p), cref<double>(d), s.tn[k], L);
            if (dir == 0) s.Lval[k] = L.v;
            s.Lgrad[k * NDER + dir] = L.d[0];
            if (NG > 0) {
                ad g[NG > 0 ? NG : 1];
                for (int q = 0; q < NG; ++q) g[q] = ad(0.0);
                model.template inequality_constraints_impl<ad>(cref<ad>(x), cref<ad>(u), cref<ad>(p), cref<double>(d), s.tn[k], vref<ad>(g));
                for (int q = 0; q < NG; ++q) {
                    if (dir == 0) s.gval[k * NG + q] = g[q].v;
                    s.gjac[(k * NG + q) * NDER + dir] = g[q].d[0];
                }
            }
            if (dir == 0) {
                int seg, row; seg_row(k, seg, row);
                // D-row times the segment's states: loads of 4 nodes at a time (independent), then the ordered adds
                double acc[NX > 0 ? NX : 1];
                for (int q = 0; q < NX; ++q) acc[q] = 0.0;
                for (int j0 = 0; j0 <= P; j0 += 4) {
                    double dv[4], xs[4][NX > 0 ? NX : 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int jj = (j0 + j <= P) ? j0 + j : 0;
                        dv[j] = s.D[row + jj * (P + 1)];
#pragma unroll
                        for (int q = 0; q < NX; ++q) xs[j][q] = var[(seg * P + jj) * NX + q];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j0 + j <= P) {
#pragma unroll
                            for (int q = 0; q < NX; ++q) acc[q] += dv[j] * xs[j][q];
                        }
                }
#pragma unroll
                for (int q = 0; q < NX; ++q) s.DX[k * NX + q] = acc[q];
            }
            if (k == 0) {
                ad M(0.0);
                model.template mayer_term_impl<ad>(cref<ad>(x), cref<ad>(u), cref<ad>(p), cref<double>(d), s.tn[0], M);
                if (dir == 0) s.Mval[0] = M.v;
                s.Mgrad[dir] = M.d[0];
            }
        }
    }

    // ---- per-node second-order stage: d2L, Mayer Hessian, and hes = -t_scale*sum lam_q d2f_q + sum lam_g d2g (:2128-2157)
    // Two lane mappings with identical arithmetic per entry. Up to WIDE_MAX_NDER derivative directions the inner dual number
    // is kept whole (one pass over NN*NDER lanes); beyond that one lane per Hessian entry — the wide variant's AD arrays then
    // live in private memory (7.9 KB per lane for 16 directions) and that kernel proved fragile (compiler hazard 8).
    // (forced inline: as a called member function the stage drags the whole Ocp object into private memory — hazard 6)
    static constexpr int WIDE_MAX_NDER = 8;
    __device__ __forceinline__ void stage_second_order(const double* var, const double* lam) {
        if constexpr ((int)NDER <= WIDE_MAX_NDER) stage_second_order_wide(var, lam); else stage_second_order_entry(var, lam);
    }
    // variant for few derivative directions: one lane per (node, outer direction), the inner dual carries all NDER components
    __device__ __forceinline__ void stage_second_order_wide(const double* var, const double* lam) {
        // one lane per (node, outer seed direction): the NDER directional passes of a node are independent, so they run
        // side by side on NN*NDER lanes instead of one after the other on NN lanes (same arithmetic per entry)
        for (int kd = lane_id(); kd < dm.NN * NDER; kd += WAVE) {
            const int k = kd % dm.NN, dir = kd / dm.NN;
            {
                ad2c x[NX > 0 ? NX : 1], u[NU > 0 ? NU : 1], p[NP > 0 ? NP : 1], y[NX > 0 ? NX : 1];
                seed2(var, k, dir, x, u, p);
                ad2c L(0.0);
                model.template lagrange_term_impl<ad2c>(cref<ad2c>(x), cref<ad2c>(u), cref<ad2c>(p), cref<double>(d), s.tn[k], L);
                // hes.col(dir) = L.d[dir].d  => hes(r, dir)
                for (int r = 0; r < NDER; ++r) s.Lhes[(k * NDER + dir) * NDER + r] = L.d[0].d[r];
                for (int q = 0; q < NX; ++q) y[q] = ad2c(0.0);
                ad2c tk(s.tn[k]);
                model.template dynamics_impl<ad2c>(cref<ad2c>(x), cref<ad2c>(u), cref<ad2c>(p), cref<double>(d), tk, vref<ad2c>(y));
                double col[NDER];
                for (int r = 0; r < NDER; ++r) col[r] = 0.0;
                for (int q = 0; q < NX; ++q) {
                    const double coeff = -lam[q + k * NX] * ts;
                    for (int r = 0; r < NDER; ++r) col[r] += coeff * y[q].d[0].d[r];
                }
                if (NG > 0) {
                    ad2c g[NG > 0 ? NG : 1];
                    for (int q = 0; q < NG; ++q) g[q] = ad2c(0.0);
                    model.template inequality_constraints_impl<ad2c>(cref<ad2c>(x), cref<ad2c>(u), cref<ad2c>(p), cref<double>(d), s.tn[k], vref<ad2c>(g));
                    for (int q = 0; q < NG; ++q) {
                        const double coeff = lam[q + k * NG + dm.me];
                        for (int r = 0; r < NDER; ++r) col[r] += coeff * g[q].d[0].d[r];
                    }
                }
                for (int r = 0; r < NDER; ++r) s.dhes[(k * NDER + dir) * NDER + r] = col[r];
                if (k == 0) {
                    ad2c M(0.0);
                    model.template mayer_term_impl<ad2c>(cref<ad2c>(x), cref<ad2c>(u), cref<ad2c>(p), cref<double>(d), s.tn[0], M);
                    for (int r = 0; r < NDER; ++r) s.Mhes[dir * NDER + r] = M.d[0].d[r];
                }
            }
        }
        wsync();
    }

    // ---- per-node second-order stage: d2L, Mayer Hessian, and hes = -t_scale*sum lam_q d2f_q + sum lam_g d2g (:2128-2157)
    // One lane per Hessian ENTRY (node k, outer seed direction dir, inner derivative component r): the inner components of a
    // nested dual number never mix, so each lane carries Dual<Dual<double,1>,1> (4 doubles per AD variable instead of
    // 2*(NDER+1): no private-memory arrays, a small register footprint even for 16 derivative directions) and every entry
    // goes through exactly the operations it went through as component r of the wide inner dual.
    __device__ __forceinline__ void stage_second_order_entry(const double* var, const double* lam) { stage_second_order_entry_part<1>(var, lam, 0); wsync(); }
    // the entries e = w * 64 + lane, stride 64 NW: wavefront w of a team of NW (the large-instance kernel on a workgroup of four wavefronts, BigTeam): every
    // entry is independent of every other one; the caller synchronises the team behind it
    template <int NW>
    __device__ __forceinline__ void stage_second_order_entry_part(const double* var, const double* lam, int w) {
        using adi = Dual<double, 1>;
        using ad2 = Dual<adi, 1>;
        for (int e = lane_id() + WAVE * w; e < dm.NN * NDER * NDER; e += WAVE * NW) {
            const int k = e % dm.NN, dr = e / dm.NN, dir = dr / NDER, r = dr - dir * NDER;
            ad2 y[NX > 0 ? NX : 1];
            // second-order seeding (continuous_ocp.hpp:691-735 restricted to one outer and one inner partial), generated element by element where the
            // model reads it (see the vref specialisation in pmpc_models.hpp)
            const cref<ad2> x(var + k * NX, 0, r, dir), u(var + dm.VARX + k * NU, NX, r, dir), p(var + dm.VARX + dm.VARU, NX + NU, r, dir);
            ad2 L(0.0);
            model.template lagrange_term_impl<ad2>(x, u, p, cref<double>(d), s.tn[k], L);
            // hes.col(dir) = L.d[dir].d  => hes(r, dir)
            s.Lhes[(k * NDER + dir) * NDER + r] = L.d[0].d[0];
            for (int q = 0; q < NX; ++q) y[q] = ad2(0.0);
            ad2 tk(s.tn[k]);
            model.template dynamics_impl<ad2>(x, u, p, cref<double>(d), tk, vref<ad2>(y));
            double col = 0.0;
            for (int q = 0; q < NX; ++q) {
                const double coeff = -lam[q + k * NX] * ts;
                col += coeff * y[q].d[0].d[0];
            }
            if (NG > 0) {
                ad2 g[NG > 0 ? NG : 1];
                for (int q = 0; q < NG; ++q) g[q] = ad2(0.0);
                model.template inequality_constraints_impl<ad2>(x, u, p, cref<double>(d), s.tn[k], vref<ad2>(g));
                for (int q = 0; q < NG; ++q) {
                    const double coeff = lam[q + k * NG + dm.me];
                    col += coeff * g[q].d[0].d[0];
                }
            }
            s.dhes[(k * NDER + dir) * NDER + r] = col;
            if (k == 0) {
                ad2 M(0.0);
                model.template mayer_term_impl<ad2>(x, u, p, cref<double>(d), s.tn[0], M);
                s.Mhes[dir * NDER + r] = M.d[0].d[0];
            }
        }
    }

    // ---- assemble c (m), Jacobian J (m x n column-major in HBM), cost value and cost gradient from the first-order stage
    // equalities_linearised :797-878, _inequalities_linearised_dense :546-575, cost_gradient :1210-1249
    // J(r, col) = J[r + col * ldj] (ldj = m for a plain m x n Jacobian; n+m when J is the lower block of the stacked [H;J] workspace)
    // structure = false: J already holds a linearisation of THIS problem — its zeros and its differentiation-matrix entries
    // do not depend on the iterate, so only the per-node blocks (D self entry - t_scale*df, dg) are rewritten.
    // DENSEJ = false (block-structured kernels, pmpc_qp_schur.hpp): J is never materialised — only the per-node blocks (jblk / gblk) are written
    template <bool WANT_COST = true, bool DENSEJ = true>
    __device__ __forceinline__ double assemble_first_order(double* c, double* __restrict__ J, double* cost_grad, int ldj, bool structure = true) {
        const int ln = lane_id();
        const int n = dm.n, m = dm.m;
        if (DENSEJ && structure) {
            for (int e = ln; e < m * n; e += WAVE) J[(e % m) + (size_t)(e / m) * ldj] = 0.0;
            wfence();
            wsync();
            for (int k = ln; k < dm.NN; k += WAVE) {
                int seg, row; seg_row(k, seg, row);
                for (int q = 0; q < NX; ++q) {
                    const int r = k * NX + q;
                    if (k < dm.NN - 1) {
                        for (int j = 0; j <= P; ++j) J[r + (size_t)((seg * P + j) * NX + q) * ldj] = s.D[row + j * (P + 1)] * 1.0;
                    } else {  // last node row = -reverse(first block row) (:845-846)
                        for (int j = 0; j <= P; ++j) J[r + (size_t)(dm.VARX - NX * (P + 1) + j * NX + q) * ldj] = -s.D[0 + (P - j) * (P + 1)];
                    }
                }
            }
            wfence();
            wsync();
        }
        // own-node blocks, one lane per ENTRY (node k, state row q, derivative column i):
        // J(r, own block) = [D entry on the node's own state column | 0] - t_scale * df: one store per entry, no read-modify-write
        // round trips (same arithmetic as "= D(i,j)*I" followed by "-= t_scale*jac", :824,:870-872)
        for (int e = ln; e < dm.NN * NX * NDER; e += WAVE) {
            const int k = e / (NX * NDER), rem = e - k * (NX * NDER), q = rem / NDER, i = rem - q * NDER;
            double v = (i == q) ? s.nd[k] : 0.0;
            v -= ts * s.fjac[e];
            if constexpr (DENSEJ) J[(k * NX + q) + (size_t)dm.gidx(k, i) * ldj] = v;
            if (keep_blk) jblk[(k * NX + q) * Dm::JBS + i] = v;
        }
        for (int r = ln; r < dm.me; r += WAVE) {
            double cv = -ts * s.fval[r];
            cv += s.DX[r];
            c[r] = cv;
        }
        if (NG > 0) {
            for (int e = ln; e < dm.NN * NG * NDER; e += WAVE) {
                const int kq = e / NDER, i = e - kq * NDER, k = kq / NG;
                if constexpr (DENSEJ) J[(dm.me + kq) + (size_t)dm.gidx(k, i) * ldj] = s.gjac[e];
                if (keep_blk) gblk[kq * Dm::JBS + i] = s.gjac[e];
            }
            for (int r = ln; r < dm.mi; r += WAVE) c[dm.me + r] = s.gval[r];
        }
        // cost gradient, x/u parts, one lane per (node, variable): contributions in the reference's loop order (segment s-1 as
        // node P, then segment s as node 0)
        for (int e = ln; e < dm.NN * (NX + NU); e += WAVE) {
            const int k = e / (NX + NU), i = e - k * (NX + NU);
            double gacc = 0.0;
            const int fl = s.nsr[k];
            const double lg = s.Lgrad[k * NDER + i];
            if (fl & (1 << 30)) gacc += s.nw1[k] * lg;
            if (fl & (1 << 29)) gacc += s.nw2[k] * lg;
            if (k == 0) gacc += s.Mgrad[i];
            cost_grad[dm.gidx(k, i)] = gacc;
        }
        // p-part of the cost gradient: sequential over (segment, node) as in the reference, then Mayer
        if constexpr (NP > 0) {
            for (int i = ln; i < NP; i += WAVE) {
                double a = 0.0;
                for (int sg = 0; sg < S; ++sg)
                    for (int k = 0; k <= P; ++k) a += (ts * s.w[k]) * s.Lgrad[(sg * P + k) * NDER + NX + NU + i];
                a += s.Mgrad[NX + NU + i];
                cost_grad[dm.VARX + dm.VARU + i] = a;
            }
        }
        double cst = 0.0;
        if constexpr (WANT_COST) {
            for (int sg = 0; sg < S; ++sg)
                for (int k = 0; k <= P; ++k) cst += ts * s.w[k] * s.Lval[sg * P + k];
            cst += s.Mval[0];
        }
        wfence();
        wsync();
        return cst;
    }

    // ---- the same Hessian as its per-node blocks only (NP = 0: H IS block diagonal): hb[k NDER^2 + i NDER + r] = H(gidx(k, r), gidx(k, i)) — the entries
    // assemble_hessian writes, through the same operations; block-structured kernels keep them in LDS (pmpc_qp_schur.hpp)
    __device__ __forceinline__ void assemble_hessian_blocks(double* hb) {
        static_assert(NP == 0, "block storage of the Hessian: models without parameters");
        for (int e = lane_id(); e < dm.NN * NDER * NDER; e += WAVE) {
            const int k = e / (NDER * NDER), rem = e - k * (NDER * NDER), i = rem / NDER, r = rem - i * NDER;
            double a = 0.0;
            if (k % P == 0 && k > 0) a += (ts * s.w[P]) * s.Lhes[(k * NDER + i) * NDER + r];
            if (k < dm.NN - 1) a += (ts * s.w[k % P]) * s.Lhes[(k * NDER + i) * NDER + r];
            if (k == 0) a += s.Mhes[i * NDER + r];
            a += s.dhes[(k * NDER + i) * NDER + r];
            hb[e] = a;
        }
        wsync();
    }

    // ---- NP = 1: the arrow shape — node blocks hb[k NB^2 + i NB + r] (NB = NX + NU), the border row with the corner hbrd[0 .. n) = H(p, .) and the border
    // column hbrd[n .. 2n - 1) = H(., p) — the entries assemble_hessian writes, through the same operations (corner: the sequential chain; quirk Q4 lands on H(p, 0))
    __device__ __forceinline__ void assemble_hessian_arrow(double* hb, double* hbrd) {
        static_assert(NP == 1, "arrow storage of the Hessian: one parameter");
        constexpr int NB = NX + NU;
        const int n = dm.n;
        double* hr = hbrd; double* hc = hbrd + n;
        for (int e = lane_id(); e < dm.NN * NDER * NDER; e += WAVE) {
            const int k = e / (NDER * NDER), rem = e - k * (NDER * NDER), i = rem / NDER, r = rem - i * NDER;
            if (r >= NB && i >= NB) continue;
            double a = 0.0;
            if (k % P == 0 && k > 0) a += (ts * s.w[P]) * s.Lhes[(k * NDER + i) * NDER + r];
            if (k < dm.NN - 1) a += (ts * s.w[k % P]) * s.Lhes[(k * NDER + i) * NDER + r];
            if (k == 0) a += s.Mhes[i * NDER + r];
            a += s.dhes[(k * NDER + i) * NDER + r];
            if (r < NB && i < NB) hb[k * NB * NB + i * NB + r] = a;
            else if (r >= NB) hr[dm.gidx(k, i)] = a;   // row p, column gidx(k, i)
            else hc[dm.gidx(k, r)] = a;                // row gidx(k, r), column p
        }
        wsync();
        if (lane_id() == 0) {
            double a = 0.0;
            for (int sg = 0; sg < S; ++sg)
                for (int k = 0; k <= P; ++k) a += (ts * s.w[k]) * s.Lhes[((sg * P + k) * NDER + NB) * NDER + NB];
            for (int k = 0; k < dm.NN; ++k) a += s.dhes[(k * NDER + NB) * NDER + NB];
            hr[n - 1] = a;
            hr[0] += s.Mhes[0 * NDER + (NDER - 1)];
        }
        wsync();
    }

    // ---- assemble the Lagrangian Hessian H (n x n column-major in HBM) from the second-order stage
    // cost_gradient_hessian :1256-1367 (+ quirk Q4) and the lam-weighted blocks of :2128-2173
    __device__ __forceinline__ void assemble_hessian(double* __restrict__ H, int ldh) {
        const int ln = lane_id();
        const int n = dm.n;
        for (int e = ln; e < n * n; e += WAVE) H[(e % n) + (size_t)(e / n) * ldh] = 0.0;
        wfence();
        wsync();
        // blocks that do not involve the (p,p) corner are private to their node
        for (int k = ln; k < dm.NN; k += WAVE) {
            for (int i = 0; i < NDER; ++i)
                for (int r = 0; r < NDER; ++r) {
                    if (r >= NX + NU && i >= NX + NU) continue;
                    double a = 0.0;
                    if (k % P == 0 && k > 0) a += (ts * s.w[P]) * s.Lhes[(k * NDER + i) * NDER + r];
                    if (k < dm.NN - 1) a += (ts * s.w[k % P]) * s.Lhes[(k * NDER + i) * NDER + r];
                    if (k == 0) a += s.Mhes[i * NDER + r];
                    a += s.dhes[(k * NDER + i) * NDER + r];
                    H[dm.gidx(k, r) + (size_t)dm.gidx(k, i) * ldh] = a;
                }
        }
        if constexpr (NP > 0) {
            wfence();
            wsync();
            // (p,p) corner: sequential accumulation in reference order; Mayer pp-block lands in the bottom-LEFT corner (Q4)
            for (int e = ln; e < NP * NP; e += WAVE) {
                const int r = e % NP, i = e / NP;
                double a = 0.0;
                for (int sg = 0; sg < S; ++sg)
                    for (int k = 0; k <= P; ++k) a += (ts * s.w[k]) * s.Lhes[((sg * P + k) * NDER + NX + NU + i) * NDER + NX + NU + r];
                for (int k = 0; k < dm.NN; ++k) a += s.dhes[(k * NDER + NX + NU + i) * NDER + NX + NU + r];
                H[(n - NP + r) + (size_t)(n - NP + i) * ldh] = a;
            }
            wfence();
            wsync();
            for (int e = ln; e < NP * NP; e += WAVE) {
                const int a_ = e % NP, b_ = e / NP;
                H[(n - NP + a_) + (size_t)b_ * ldh] += s.Mhes[b_ * NDER + (NDER - NP + a_)];
            }
        }
        wfence();
        wsync();
    }
};

}  // namespace pmpc
