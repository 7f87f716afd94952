"""BASELINE configs[4] on however many GPUs the box shows: one rank per GPU (torch.distributed.run,
backend nccl = RCCL), streams sharded s -> rank s mod N, each rank's HIP bank codes its streams and
the compressed sizes are gathered from DEVICE tensors.  Also bench.py's own N > 1 plumbing with the
nccl backend.  (The 8-GPU run itself is the driver's; with one GPU visible this is world size 1.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _n_gpus():
    import torch
    return min(4, torch.cuda.device_count())  # counting devices does not initialise HIP; <= 6 GPU processes allowed


@pytest.mark.gpu
def test_compressed_sizes_gathered_over_rccl(gpu, oracle, tmp_path):
    from gmix_amd import topology
    n = _n_gpus()
    assert n >= 1
    n_streams = 5
    out = tmp_path / "gather.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "helpers", "nccl_gather.py"), str(n_streams), str(out)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.load(open(out))
    assert got["world"] == n and got["backend"] == "nccl"
    topo = topology.stock(90)
    expect = []
    for s in range(n_streams):
        pred, act, ctx, bits = oracle.synth(90, 33, 256, seed=500 + s, ctx_mode=3, ctx_mod=6, zero_mod=8, bit_mode=1)
        p, _ = oracle.Bank(90, topo.skip, topo.mixers).run(pred, act, ctx, bits, want_all=False)
        expect.append(len(oracle.encode(bits, p)))
    assert got["sizes"] == expect          # GPU banks on every rank == the oracle, gathered on every rank
    assert got["slowest"] == float(n) and got["total"] == n_streams


@pytest.mark.gpu
@pytest.mark.report
def test_bench_multi_rank_path_over_rccl(gpu):
    """bench.py's N > 1 code path (init nccl, barrier, MAX/SUM all-reduce) with one rank per visible GPU."""
    n = _n_gpus()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GMX_BENCH_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    # started plainly: for n > 1 bench.py itself spawns the ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--streams", "256", "--bits", "128",
                        "--steps", "3", "--warmup", "1", "--no-also", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n and line["config"]["bits_per_step"] == 256 * 128 * n
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


@pytest.mark.gpu
@pytest.mark.report
def test_bench_whole_files_on_every_rank(gpu):
    """bench.py across ranks also compresses whole files on every GPU (64 files per rank on the rank's own device, a
    process per GPU, nothing exchanged but the figures): also.e2e_S64 carries every rank's rate and the aggregate."""
    n = _n_gpus()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GMX_BENCH_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--streams", "256", "--bits", "128",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--only-also", "e2e_S64", "--e2e-many-bytes", "6000"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    e = line["also"]["e2e_S64"]
    assert "error" not in e, e
    assert e["n_gpus"] == n and len(e["per_rank"]["bits_per_s"]) == n and min(e["per_rank"]["bits_per_s"]) > 1e5
    assert e["identical_to_stock"] is True and e["value"] > 1e5
