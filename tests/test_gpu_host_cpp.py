"""The C++ host mirrors of the reference's plug-in surface (gmix_amd/host/gmx_mixer.h: Mixer /
Predictor slice; gmx_models.h: Indirect, LstmModel): compiled with g++ against libgmxmix.so and run
on the GPU box."""
import os
import subprocess

import numpy as np
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


def test_cpp_indirect_and_lstm_models_match_oracle(gpu, oracle, tmp_path):
    z = np.load(os.path.join(ROOT, "tests", "golden", "ind_tiny_dense.npz"))   # the two next-state tables
    z["ns_next"].astype(np.uint8).tofile(str(tmp_path / "ns_next.bin"))
    z["rm_next"].astype(np.uint8).tofile(str(tmp_path / "rm_next.bin"))
    exe = str(tmp_path / "test_host_models")
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_host_models.cpp"),
        "-L" + os.path.join(ROOT, "gmix_amd"), "-lgmxmix", "-L" + os.path.join(ROOT, "oracle"), "-lgmxoracle",
        "-Wl,-rpath," + os.path.join(ROOT, "gmix_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
        "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "Tests passed." in out.stdout, out.stdout + out.stderr
