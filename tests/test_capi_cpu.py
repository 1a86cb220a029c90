"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol include/polympc_amd.h declares,
refuses to run without a GPU (no fallback), and its host-side collocation constants match the pinned oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pa():
    import polympc_amd
    polympc_amd.build_library()
    return polympc_amd


def test_header_symbols_are_exported(pa):
    hdr = open(os.path.join(ROOT, "include", "polympc_amd.h")).read()
    hdr = re.sub(r"typedef[^;]*;", "", hdr)          # function-pointer typedefs are types, not exports
    declared = sorted(set(re.findall(r"\b(pmpc_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    assert sorted(pa.EXPORTED_SYMBOLS) == declared
    lib = pa.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/polympc_amd.h but not exported"


def test_struct_layouts_match_header(pa):
    import ctypes as C
    assert C.sizeof(pa.QPSettings) == 72 and C.sizeof(pa.QPInfo) == 40
    assert C.sizeof(pa.SQPSettings) == 112 and C.sizeof(pa.SQPInfo) == 48


def test_defaults_match_reference(pa):
    q = pa.qp_settings_default()      # qp_base.hpp:17-53
    assert (q.eps_rel, q.eps_abs, q.max_iter, q.rho, q.sigma, q.alpha) == (1e-3, 1e-3, 1000, 1e-1, 1e-6, 1.0)
    assert (q.check_termination, q.adaptive_rho, q.adaptive_rho_tolerance, q.adaptive_rho_interval) == (25, 0, 5.0, 25)
    s = pa.qp_settings_sqp_default()  # sqp_base.hpp:83-90
    assert (s.check_termination, s.eps_abs, s.eps_rel, s.max_iter, s.adaptive_rho, s.adaptive_rho_interval, s.alpha) == (10, 1e-4, 1e-4, 100, 1, 50, 1.0)
    n = pa.sqp_settings_default()     # sqp_base.hpp:24-47
    assert (n.tau, n.eta, n.rho, n.eps_prim, n.eps_dual, n.max_iter, n.line_search_max_iter) == (0.5, 0.25, 0.5, 1e-3, 1e-3, 100, 100)
    assert (n.line_search, n.filter_max_depth, n.filter_beta, n.filter_state) == (0, 10, 1e-5, None)   # LSFilter, line_search.hpp:38-39
    assert (n.iteration_trace, n.iteration_trace_capacity) == (None, 0)   # iteration_callback = nullptr, sqp_base.hpp:33


def test_no_cpu_fallback(pa):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        pa.Context(0)


@pytest.mark.parametrize("P", [2, 3, 5, 6, 10, 15])
def test_host_chebyshev_matches_oracle(pa, oracle, P):
    n1, w1, D1 = pa.chebyshev(P)
    n2, w2, D2 = oracle.cheb(P)
    assert np.array_equal(n1, n2) and np.array_equal(w1, w2) and np.array_equal(D1, D2)


@pytest.mark.parametrize("model", [0, 1, 2, 3, 4])
def test_dims_match_oracle(pa, oracle, model):
    for P, S in ((6, 1), (5, 2), (5, 3), (3, 2)):
        assert pa.ocp_dims(model, P, S) == oracle.ocp_dims(model, P, S)


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under polympc_amd/ or include/ may import, include or link it."""
    for base in ("polympc_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    bad = re.search(r'#include\s*[<"][^>"]*oracle|import\s+oracle|from\s+oracle|liboracle|oracle/', txt)
                    assert bad is None, f"{os.path.join(dp, f)} reaches into the oracle: {bad.group(0)}"


def test_workload_generator_is_deterministic():
    from polympc_amd import workloads
    a = workloads.robot_batch(8); b = workloads.robot_batch(4, first=4)
    assert np.array_equal(a["lbx"][4:], b["lbx"]) and np.array_equal(a["d"][4:], b["d"])
    u = workloads.uniform_pm1(workloads.SEED, np.arange(1000), 0)
    assert -1 <= u.min() < -0.9 and 0.9 < u.max() <= 1 and abs(u.mean()) < 0.1


def test_bench_rejects_a_launcher_mismatch():
    """bench.py --gpus N under a launcher that started a different number of ranks is an error (never a silent single-GPU run)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_argument_validation_without_a_device():
    """Entry points reject bad arguments before touching the GPU (so this runs on a CPU box): NULL context, and — with any non-NULL
    context pointer never dereferenced on these paths — max_iter < 1 and a missing static-parameter array for a model with ND > 0."""
    import ctypes as C
    import polympc_amd as pa
    L = pa.lib()
    ss = pa.sqp_settings_default(); qs = pa.qp_settings_sqp_default()
    one = (C.c_double * 64)()
    info = (C.c_char * 48)()
    f = L.pmpc_sqp_solve_batch_dev
    f.restype = C.c_int
    args = lambda ctx, d, s: (ctx, 0, 6, 1, C.c_double(0.0), C.c_double(2.0), None, 0, 1, None, None, d, one, one, None, None, C.byref(s), C.byref(qs), one, one, info)
    fake = C.c_void_p(8)   # never dereferenced: validation precedes every use of the context
    assert f(*args(None, one, ss)) == 1                       # PMPC_ERR_INVALID_ARGUMENT: NULL context
    assert f(*args(fake, None, ss)) == 1                      # ND = 1 (robot) and d == NULL
    s0 = pa.sqp_settings_default(); s0.max_iter = 0
    assert f(*args(fake, one, s0)) == 1                       # nothing would be solved
