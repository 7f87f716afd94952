"""The product's scalar math (gmix_amd/csrc/gmx_math.h, host compile) against the libm the
reference links (sigmoid.cpp:5): exhaustive over all 2^32 float inputs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def mc():
    src = os.path.join(HERE, "helpers", "mathcheck.c")
    so = os.path.join(HERE, "helpers", "libmathcheck.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "gmix_amd", "csrc", "gmx_math.h"))):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", src,
                               "-o", so, "-lm"])
    L = C.CDLL(so)
    for f in (L.gmx_check_expf_range, L.gmx_check_logistic_range, L.gmx_check_logf_range,
              L.gmx_check_expm1f_range, L.gmx_check_tanhf_range):
        f.restype = C.c_uint64
        f.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
    L.gmx_host_logistic_array.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.gmx_host_squash_array.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    return L


@pytest.mark.slow
def test_expf_equals_libm_everywhere(mc):
    bad = np.zeros(16, np.uint32)
    n = mc.gmx_check_expf_range(0, 0xFFFFFFFF, bad.ctypes.data, 16)
    assert n == 0, [hex(b) for b in bad[:min(n, 16)]]


@pytest.mark.slow
def test_logistic_equals_reference_everywhere(mc):
    bad = np.zeros(16, np.uint32)
    n = mc.gmx_check_logistic_range(0, 0xFFFFFFFF, bad.ctypes.data, 16)
    assert n == 0, [hex(b) for b in bad[:min(n, 16)]]


def test_squash_clamp_matches_oracle(mc, oracle):
    x = np.concatenate([np.linspace(-30, 30, 20001), [0.0, -0.0, 9.21, -9.21, 9.2103, 88, -104, 1e30, -1e30]]).astype(np.float32)
    y = np.zeros_like(x)
    mc.gmx_host_squash_array(x.ctypes.data, y.ctypes.data, len(x))
    ref = np.array([oracle.lib().gmxo_squash_clamp(float(v)) for v in x], np.float32)
    assert np.array_equal(y.view(np.uint32), ref.view(np.uint32))
    assert y.min() == np.float32(0.0001) and y.max() == np.float32(1) - np.float32(0.0001)


@pytest.mark.slow
@pytest.mark.parametrize("fn", ["logf", "expm1f", "tanhf"])
def test_lstm_math_equals_libm_everywhere(mc, fn):
    """gmx_logf / gmx_expm1f / gmx_tanhf (what the LSTM byte model calls through Sigmoid::Logit,
    tanh(float) and tanh(valarray)) against the machine's libm for every float."""
    bad = np.zeros(16, np.uint32)
    n = getattr(mc, f"gmx_check_{fn}_range")(0, 0xFFFFFFFF, bad.ctypes.data, 16)
    assert n == 0, (n, [hex(b) for b in bad[:min(n, 16)]])
