import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    """A HIP device is visible — asked of the HIP runtime itself (hipGetDeviceCount through ctypes): the product is a ctypes library and does
    not need torch, so a box without torch (or with a CPU-only wheel) must not turn the GPU suite into a green run of skips."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
            n = ctypes.c_int(0)
            if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
                return True
            return False
        except OSError:
            continue
    return False


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a HIP device: on a box without one they are skipped (a plain `pytest` then runs the CPU suite instead of
    erroring in the Context fixture). On a GPU box nothing is skipped — and the product fails loudly if its library is missing.
    PMPC_REQUIRE_GPU=1 (for a GPU CI driver) turns the skip into an error."""
    if _gpu_available():
        return
    if os.environ.get("PMPC_REQUIRE_GPU") == "1" and any("gpu" in item.keywords for item in items):
        raise pytest.UsageError("PMPC_REQUIRE_GPU=1 but no HIP device is visible: the gpu-marked tests would all be skipped")
    skip = pytest.mark.skip(reason="no HIP device visible (polympc_amd has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference algorithm (test infrastructure, never the product path)."""
    from oracle import binding
    binding.build()
    return binding
