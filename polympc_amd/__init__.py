"""polympc_amd — MI355X-native batched SQP / box-ADMM engine behind PolyMPC's Solver<OCP>::solve() surface.

The product is the C-ABI shared library ``polympc_amd/libpolympc_amd.so`` (HIP, gfx950) declared in
``include/polympc_amd.h``; the C++ host mirror of the reference's interface lives in ``include/polympc/``.
This Python package is only plumbing for tests and the benchmark: a ctypes binding of the C ABI that takes numpy
arrays (host entry points) or torch CUDA tensors (device entry points). There is no CPU fallback: loading fails
loudly when the HIP library is missing, and context creation fails when no GPU is visible.
"""
from .capi import (  # noqa: F401
    LIB_PATH, Context, QPSettings, QPInfo, SQPSettings, SQPInfo, build_library, chebyshev, lib, ocp_dims,
    qp_settings_default, qp_settings_sqp_default, sqp_settings_default,
    MODEL_ROBOT, MODEL_CSTR, MODEL_PARKING, MODEL_ROBOT_NG, MODEL_KITE_STANDIN, MODEL_PARKING_NG,
    QP_SOLVED, QP_MAX_ITER_EXCEEDED, SQP_SOLVED, SQP_MAX_ITER_EXCEEDED, EXPORTED_SYMBOLS,
)
