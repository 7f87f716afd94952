"""The C++ host mirror of the reference's Mixer / Predictor surface (gmix_amd/host/gmx_mixer.h):
compiled with g++ against libgmxmix.so and run on the GPU box."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_adapter_matches_oracle_and_tester_invariants(gpu, oracle, tmp_path):
    exe = str(tmp_path / "test_host_adapter")
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_host_adapter.cpp"),
        "-L" + os.path.join(ROOT, "gmix_amd"), "-lgmxmix", "-L" + os.path.join(ROOT, "oracle"), "-lgmxoracle",
        "-Wl,-rpath," + os.path.join(ROOT, "gmix_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
        "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "Tests passed." in out.stdout, out.stdout + out.stderr
