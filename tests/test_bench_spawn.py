"""`python bench.py --gpus N` started plainly must run N ranks (it used to run one and
print n_gpus: 1).  On CPU: the parent spawns torch.distributed.run as a child; --rehearse-cpu makes
the ranks meet over gloo and skip the GPU work, so only the launch plumbing is exercised here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=600)


def test_plain_start_spawns_n_ranks():
    r = _run(["--gpus", "2", "--rehearse-cpu"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_counted"] == 2 and line["max_rank"] == 1.0


def test_eight_ranks_rehearsal():
    """The 8-GPU launch of BASELINE configs[4], rehearsed over gloo: eight ranks meet, every collective of the bench
    runs (barrier, MAX of the elapsed time, SUM of the stream counts, the per-rank gather and MIN), rank 0 alone
    prints.  No GPU work, no figure."""
    r = _run(["--gpus", "8", "--rehearse-cpu"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks_counted"] == 8 and line["max_rank"] == 7.0
    assert line["per_rank"] == [100.0 + k for k in range(8)] and line["min"] == 100.0


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--rehearse-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()
