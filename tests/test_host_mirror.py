"""The C++ host-side mirror of the reference's interface (include/polympc/polympc.hpp) and the user-OCP registration path
(include/polympc/register_ocp.hpp), exercised by tests/cpp/host_mirror_test.cpp — written like the reference's gtest cases."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")
BIN = os.path.join(CPP, "host_mirror_test")


def _build():
    import polympc_amd
    polympc_amd.build_library()
    subprocess.check_call(["make", "-C", CPP, "-s"])


def test_host_mirror_builds_and_refuses_to_run_without_gpu():
    """g++ compiles the header-only mirror (no Eigen, no HIP headers needed on the host side); hipcc compiles the user's
    OCP translation unit. Without a GPU the binary must exit 77 (no CPU fallback)."""
    import torch
    _build()
    assert os.path.exists(BIN) and os.path.exists(os.path.join(CPP, "libuser_ocp.so"))
    if not torch.cuda.is_available():
        r = subprocess.run([BIN], capture_output=True, text=True)
        assert r.returncode == 77 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_host_mirror_reference_style_cases():
    if not os.path.exists(BIN):
        _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL PASSED" in r.stdout


def test_uniform_divisor_quotient_is_the_ieee_quotient():
    """The 5-operation corrected quotient the BFGS update uses for its two wave-uniform divisors (UniformDiv) equals a / b
    bit for bit: 20 M random and structured (significands next to 1 and 2) operand pairs, restated on the CPU."""
    _build()
    r = subprocess.run([os.path.join(CPP, "uniform_div_check"), "20000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr
