"""Comparison of two solution sets of one synthetic batch (numpy only — no solver arithmetic, no oracle): the record that the bench line, the GPU
parity tests, the pin tests of the CPU checker and its cross-order tool share when they put a run next to the reference-order run."""
import numpy as np


def variable_scales(cfg, wl):
    """Per-variable magnitudes (n) and per-multiplier magnitudes are NOT invented here: x scales are the steady state / box of the workload
    definitions in polympc_amd/workloads.py (CSTR: states (1, 0.5, 100, 100), inputs (35, 9000); robot: states 1, inputs (1.5, 0.75); kite
    stand-in: 1)."""
    P, S = wl["P"], wl["S"]
    nn = P * S + 1
    if cfg == "B":
        xs, us = [1.0, 0.5, 100.0, 100.0], [35.0, 9000.0]
    elif cfg == "C":
        xs, us = [1.0] * 13, [1.0] * 3
    else:
        xs, us = [1.0, 1.0, 1.0], [1.5, 0.75]
    return np.concatenate([np.tile(xs, nn), np.tile(us, nn)])


def cross_order_stats(cfg, wl, x, lam, info, xr, lamr, inforef):
    """(x, lam, info) — a solution set under test (numpy arrays; info = structured array or list of oracle SQPInfo) — against the reference-order run
    (xr, lamr, inforef). Returns the record the bench line / tests / table share. Solution differences are reported over ALL instances and over the
    instances whose trajectory (SQP iterations, status, total ADMM iterations) is identical in both runs: an instance that takes another branch at a
    borderline discrete decision is a different computation, not a perturbed one."""
    def col(inf, f):
        return np.asarray(inf[f]) if isinstance(inf, np.ndarray) else np.array([getattr(i, f) for i in inf])
    B = x.shape[0]
    same = (col(info, "iter") == col(inforef, "iter")) & (col(info, "status") == col(inforef, "status")) & \
           (col(info, "qp_solver_iter") == col(inforef, "qp_solver_iter"))
    sc = variable_scales(cfg, wl)
    n = x.shape[1]
    if sc.size != n:   # models with parameters (none of the BASELINE configurations)
        sc = np.concatenate([sc, np.ones(n - sc.size)])
    dx = np.abs(x - xr)
    dxi = dx.max(axis=1)
    dxs = (dx / sc).max(axis=1)
    dl = np.abs(lam - lamr)
    lscale = np.maximum(1.0, np.abs(lamr).max(axis=1))
    dls = dl.max(axis=1) / lscale
    cost_r = col(inforef, "cost")
    dviol = np.abs(col(info, "max_violation") - col(inforef, "max_violation"))
    dcost = np.abs(col(info, "cost") - cost_r) / np.maximum(1.0, np.abs(cost_r))
    dpn = np.abs(col(info, "primal_norm") - col(inforef, "primal_norm")); ddn = np.abs(col(info, "dual_norm") - col(inforef, "dual_norm"))
    pct = lambda v: {"p50": float(np.percentile(v, 50)), "p90": float(np.percentile(v, 90)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
    mx = lambda v, msk: float(v[msk].max()) if msk.any() else 0.0
    # an instance can take another branch at a borderline decision and still end with the same counts (config B, 16 384 instances: one such instance
    # ends 1.55 apart): "unbranched" also drops the instances whose two runs ended more than 1e-4 (scaled) apart, and says how many those were
    unbr = same & (dxs <= 1e-4)
    return {
        "instances": int(B), "identical_trajectory_fraction": float(same.mean()), "different_trajectories": int((~same).sum()),
        "max_abs_dx": float(dx.max()), "instances_dx_over_1e-8": int((dxi > 1e-8).sum()),
        "abs_dx_per_instance": pct(dxi), "scaled_dx_per_instance": pct(dxs),
        "max_abs_dlam": float(dl.max()), "scaled_dlam_per_instance": pct(dls),
        "max_abs_d_primal_norm": float(dpn.max()), "max_abs_d_dual_norm": float(ddn.max()),
        "max_abs_d_constraint_violation": float(dviol.max()), "max_rel_d_cost": float(dcost.max()),
        "identical_trajectories_only": {"max_abs_dx": mx(dxi, same), "max_scaled_dx": mx(dxs, same), "max_scaled_dlam": mx(dls, same),
                                        "max_abs_d_constraint_violation": mx(dviol, same), "max_rel_d_cost": mx(dcost, same)},
        "unbranched_only": {"instances": int(unbr.sum()), "same_counts_but_over_1e-4": int((same & ~unbr).sum()), "max_scaled_dx": mx(dxs, unbr),
                            "max_scaled_dlam": mx(dls, unbr), "max_abs_d_constraint_violation": mx(dviol, unbr), "max_rel_d_cost": mx(dcost, unbr)},
        "scaling": "dx per variable / its box or steady-state magnitude (CSTR: states 1, 0.5, 100, 100, inputs 35, 9000; robot: states 1, inputs 1.5, 0.75; "
                   "stand-in: 1); dlam per instance / max(1, |lam_ref|_inf)",
    }


def qp_level_stats(x, y, info, xr, yr, inforef):
    """ONE box-ADMM solve per QP (SURVEY 8d's unit; north_star: "primal/dual KKT residual within 1e-8 of CPU reference"): a solution set under test
    (x [Q, n], y [Q, m + n], info: structured array or list with status / iter / rho_updates / res_prim / res_dual — the quantities of
    qp_base.hpp:240-252 and box_admm.hpp:398-431) against the reference-order solve of the same QPs. No masks: every QP enters."""
    def col(inf, f):
        return np.asarray(inf[f]) if isinstance(inf, np.ndarray) else np.array([getattr(i, f) for i in inf])
    pct = lambda v: {"p50": float(np.percentile(v, 50)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
    dx = np.abs(x - xr).max(axis=1); dy = np.abs(y - yr).max(axis=1)
    ys = dy / np.maximum(1.0, np.abs(yr).max(axis=1))
    drp = np.abs(col(info, "res_prim") - col(inforef, "res_prim")); drd = np.abs(col(info, "res_dual") - col(inforef, "res_dual"))
    return {
        "qps": int(x.shape[0]),
        "different_iter": int((col(info, "iter") != col(inforef, "iter")).sum()),
        "different_status": int((col(info, "status") != col(inforef, "status")).sum()),
        "different_rho_updates": int((col(info, "rho_updates") != col(inforef, "rho_updates")).sum()),
        "max_abs_d_res_prim": float(drp.max()), "max_abs_d_res_dual": float(drd.max()),
        "abs_dx_per_qp": pct(dx), "abs_dy_per_qp": pct(dy), "scaled_dy_per_qp": pct(ys),
        "mean_admm_iterations": float(col(inforef, "iter").mean()),
        "solved_fraction_reference": float((col(inforef, "status") == 0).mean()),
    }
