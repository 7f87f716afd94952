"""Device scalar math == host scalar math (the same gmx_math.h), which tests/test_math.py
shows equals the reference's libm-based Sigmoid::Logistic: exhaustive over all float inputs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_math_equals_host_math_samples(gpu, oracle):
    L = gpu._lib.lib()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.integers(0, 1 << 32, 2_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32),
                        np.linspace(-110, 95, 400001).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 88.7228, -103.9721, -87.3], np.float32)])
    so = os.path.join(HERE, "helpers", "libmathcheck.so")
    if not os.path.exists(so):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
                               os.path.join(HERE, "helpers", "mathcheck.c"), "-o", so, "-lm"])
    H = C.CDLL(so)
    for what, hostfn in ((1, H.gmx_host_logistic_array), (2, H.gmx_host_squash_array)):
        y = np.zeros_like(x)
        assert L.gmx_debug_math_probe(0, x.ctypes.data, y.ctypes.data, len(x), what) == 0
        ref = np.zeros_like(x)
        hostfn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        hostfn(x.ctypes.data, ref.ctypes.data, len(x))
        same = (y.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(y) & np.isnan(ref))
        assert same.all(), x[~same][:5]
    # expf against libm directly
    y = np.zeros_like(x)
    assert L.gmx_debug_math_probe(0, x.ctypes.data, y.ctypes.data, len(x), 0) == 0
    ref = np.zeros_like(x)
    oracle.lib().gmxo_libm_expf_array(x.ctypes.data, ref.ctypes.data, len(x))
    same = (y.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(y) & np.isnan(ref))
    assert same.all(), x[~same][:5]


def test_device_logistic_exhaustive_checksum(gpu):
    """All 2^32 inputs: the device folds its results into a checksum; the host does the same
    with its own (libm-pinned) implementation."""
    L = gpu._lib.lib()
    src = os.path.join(HERE, "helpers", "rangecheck.c")
    so = os.path.join(HERE, "helpers", "librangecheck.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", src, "-o", so, "-lm"])
    H = C.CDLL(so)
    H.gmx_host_math_range.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_ulonglong)]
    for what in (0, 1, 2):
        dev = (C.c_ulonglong * 2)()
        host = (C.c_ulonglong * 2)()
        assert L.gmx_debug_math_range(0, 0, 1 << 32, what, dev) == 0
        H.gmx_host_math_range(0, 1 << 32, what, host)
        assert (dev[0], dev[1]) == (host[0], host[1]), what


def test_wave_level_short_ways_equal_the_general_expressions(gpu):
    """gmx_stock.hip's logistic and row-age division take shorter instruction sequences when a whole wave is in
    range (gmx_math.h, gmx_wave_*): the logistic over all 2^32 inputs folds to the same checksum as the general
    function (which the test above pins to libm), and the short double division equals the compiler's on 4096 x 4096
    small counter pairs and 2^32 hashed ones (what = 4 counts the differing results)."""
    L = gpu._lib.lib()
    gen = (C.c_ulonglong * 2)()
    short = (C.c_ulonglong * 2)()
    assert L.gmx_debug_math_range(0, 0, 1 << 32, 1, gen) == 0
    assert L.gmx_debug_math_range(0, 0, 1 << 32, 3, short) == 0
    assert (short[0], short[1]) == (gen[0], gen[1])
    bad = (C.c_ulonglong * 2)()
    assert L.gmx_debug_math_range(0, 0, (1 << 32) + (1 << 24), 4, bad) == 0
    assert (bad[0], bad[1]) == (0, 0), "short row-age division differs on %d pairs (xor of 1 + their indices: %d)" % (bad[1], bad[0])
