import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a HIP device: on a box without one they are skipped (a plain `pytest` then runs the CPU suite instead of
    erroring in the Context fixture). On a GPU box nothing is skipped — and the product fails loudly if its library is missing."""
    if _gpu_available():
        return
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
