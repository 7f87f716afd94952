#!/usr/bin/env python3
"""End-to-end bench lines: whole files through the reference's own feature models and coder on the host cores with
the run-ahead compressor (gmix_amd/host/gmx_batched.h) taking the device-side models in batches -- BASELINE.json
configs[2] / [3] in small, on text that is on both boxes (GMX_CORPUS=/path/to/enwik8 for the named data).

  variant "mixers": the 33 mixers on the MI355X (oracle/_ref/gmix_many); the host keeps all 88 feature models
  variant "chain":  LSTM + 41 Indirect models + 33 mixers on the MI355X (gmix_chain_many); the host keeps PPMd, the
                    match models, the context hashes and the coder

S files of n_bytes each are compressed side by side (one Predictor and host thread per file, ONE device group);
`value` = bits of all files / wall time of the compression phase (Predictors built before the clock starts; the
build time is reported).  Every output is compared with the stock strict build's `gmix -c` of the same file
(`identical_to_stock`).  cpu_baseline = the reference's own CLI with the makefile's -Ofast on the same file(s),
as many processes at once as there are files (at most 16): what the same host does without the device.
  python scripts/bench_e2e.py [--streams S --bytes N --variant chain|mixers --chunk T]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
EXE = {"mixers": "gmix_many", "chain": "gmix_chain_many"}


def corpus(n_bytes, offset=0):
    path = os.environ.get("GMX_CORPUS")
    if path:
        with open(path, "rb") as f:
            f.seek(offset)
            data = f.read(n_bytes)
        if len(data) == n_bytes:
            return data, path
    data = b"".join(open(os.path.join(ROOT, f), "rb").read() for f in ("DESIGN.md", "SURVEY.md", "INTEGRATION.md"))
    while len(data) < offset + n_bytes:
        data += data
    return data[offset:offset + n_bytes], "DESIGN.md+SURVEY.md+INTEGRATION.md of this repository"


def host_cpu():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_quota():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        return None


def measure(streams=1, n_bytes=30000, variant="chain", chunk=2048, verify=True, cpu=True, device=None):
    exe = os.path.join(REF, EXE[variant])
    if not os.path.exists(exe):
        raise RuntimeError(f"{exe} missing (make -C oracle/ref_build batched, needs /root/reference)")
    S = streams
    with tempfile.TemporaryDirectory() as tmp:
        files, src = [], None
        for k in range(S):
            data, src = corpus(n_bytes, 1531 * k)
            f = os.path.join(tmp, f"f{k}")
            open(f, "wb").write(data)
            files.append(f)
        r = subprocess.run([exe, "-T", str(chunk)] + (["--device", str(device)] if device is not None else []) +
                           [os.path.join(tmp, "out")] + files, capture_output=True, text=True, timeout=1500)
        if r.returncode != 0:
            raise RuntimeError(f"{EXE[variant]} failed: {r.stderr[-500:]}")
        st = json.loads(r.stdout.strip().splitlines()[-1])
        bits = 8 * st["input_bytes"]
        quota = cpu_quota()
        out = {"metric": "whole-compressor bits/sec (reference feature models + coder on the host, "
                         + ("33 mixers" if variant == "mixers" else "LSTM + 41 Indirect models + 33 mixers")
                         + " on the MI355X in run-ahead batches)",
               "value": st["bits_per_second"], "unit": "bits/s", "n_gpus": 1, "higher_is_better": True,
               "dtype": "f32", "data": "real text",
               "config": {"workload": f"{S} file(s) x {n_bytes} bytes of {src}, compressed side by side, "
                                      f"{chunk}-bit chunks through a ring of four batches", "streams": S,
                          "bytes_per_file": n_bytes, "chunk_bits": chunk, "variant": variant,
                          "host_threads": S, "host_cpu": host_cpu(), "cpus_visible": os.cpu_count(),
                          "cpu_quota_cores": quota, "threads_pinned_to_gpu_numa_node": st["pinned_threads"]},
               "seconds": st["wall_seconds"], "us_per_bit_per_stream": st["wall_seconds"] * 1e6 / (8 * n_bytes),
               "predictor_build_seconds": st["build_seconds"], "launches": st["launches"],
               # the submitting thread's time queueing chunks / waiting for the chunk before: while it waits, the
               # device (not the hosts' feature models) sets the pace
               "device_submit_frac": st["submit_seconds"] / st["wall_seconds"],
               "device_wait_frac": st["wait_seconds"] / st["wall_seconds"],
               "compressed_bytes": st["output_bytes"]}
        strict = os.path.join(REF, "gmix_strict")
        if verify and os.path.exists(strict):
            def one(k):
                d = os.path.join(tmp, f"s{k}")
                os.mkdir(d)
                subprocess.run([strict, "-c", files[k], os.path.join(d, "c")], cwd=d, capture_output=True, timeout=1500,
                               check=True)
                return open(os.path.join(d, "c"), "rb").read() == open(os.path.join(tmp, "out", f"{k}.gmix"), "rb").read()
            check = list(range(S)) if S <= 16 else sorted({0, 1, S // 3, S // 2, S - 2, S - 1})
            with ThreadPoolExecutor(min(16, len(check))) as ex:
                same = list(ex.map(one, check))
            out["identical_to_stock"] = all(same)
            out["files_compared"] = len(check)
        fast = os.path.join(REF, "gmix_fast")
        if cpu and os.path.exists(fast):
            n = min(S, 16)
            t0 = time.perf_counter()
            def run(k):
                d = os.path.join(tmp, f"c{k}")
                os.mkdir(d)
                subprocess.run([fast, "-c", files[k], os.path.join(d, "c")], cwd=d, capture_output=True, timeout=1500,
                               check=True)
            with ThreadPoolExecutor(n) as ex:
                list(ex.map(run, range(n)))
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": 8 * n_bytes * n / dt, "unit": "bits/s", "cores": n, "kind": "reference",
                                   "sample": f"{n} process(es) of the reference's own `gmix -c` at once (whole CLI: "
                                             f"Predictor construction included), -Ofast -march=x86-64-v3 (the makefile's "
                                             f"flags with a portable -march, not native), {n_bytes} bytes each, on "
                                             f"{host_cpu()}" + (f", cpu quota {quota:g} cores" if quota else "")}
    return out


def measure_training(train_bytes=20000, test_bytes=4000, variant="chain", verify=True, cpu=True):
    """`gmix -t train test` (runner_utils::RunTraining, runner-utils.cpp:222-322): the training Predictor and the copy
    scored on the test file every other per cent both run ahead of the device (gmx::BatchedRunTraining).  `value` =
    bits through a Predictor (training bits + 49 x the test bits) / wall time of the whole CLI -- the 1 + 49
    constructions and copies of a Predictor, which are the reference's own host work, included.  cpu_baseline = the
    reference's -Ofast CLI doing the same; identical_to_stock = data/tmp, analysis/training.tsv and
    data/trained_checkpoint.long against the strict build's."""
    exe = {"chain": "gmix_chain_batched", "mixers": "gmix_batched"}[variant]
    for e in (exe,):
        if not os.path.exists(os.path.join(REF, e)):
            raise RuntimeError(f"oracle/_ref/{e} missing (make -C oracle/ref_build batched, needs /root/reference)")
    with tempfile.TemporaryDirectory() as tmp:
        train, src = corpus(train_bytes, 500)
        test, _ = corpus(test_bytes, 90000)
        open(os.path.join(tmp, "train"), "wb").write(train)
        open(os.path.join(tmp, "test"), "wb").write(test)

        def run(e):
            d = os.path.join(tmp, e)
            os.mkdir(d)
            t0 = time.perf_counter()
            r = subprocess.run([os.path.join(REF, e), "-t", os.path.join(tmp, "train"), os.path.join(tmp, "test")], cwd=d,
                               capture_output=True, text=True, timeout=1500)
            dt = time.perf_counter() - t0
            if r.returncode != 0 or "training cross entropy" not in r.stdout:
                raise RuntimeError(f"{e} -t failed: {r.stdout[-300:]} {r.stderr[-500:]}")
            return dt, r.stdout[r.stdout.index("training cross entropy"):].splitlines()[0]

        dt, said = run(exe)
        percent = 1 + train_bytes // 100
        evals = sum(1 for pos in range(1, train_bytes) if pos % percent == 0 and (pos // percent) % 2 == 0)
        bits = 8 * (train_bytes + evals * test_bytes)
        out = {"metric": "training bits/sec (`gmix -t`: a Predictor trained on one file and a copy of it scored on a test "
                         "file every other per cent, device-side models in run-ahead batches)",
               "value": bits / dt, "unit": "bits/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "real text",
               "config": {"workload": f"gmix -t: {train_bytes} training bytes, {test_bytes} test bytes scored {evals} times, of {src}",
                          "streams": 1, "train_bytes": train_bytes, "test_bytes": test_bytes, "evaluations": evals,
                          "variant": variant, "host_cpu": host_cpu(), "cpu_quota_cores": cpu_quota()},
               "seconds": dt, "bits": bits, "printed": said}
        others = [e for e, want in (("gmix_strict", verify), ("gmix_fast", cpu)) if want and os.path.exists(os.path.join(REF, e))]
        with ThreadPoolExecutor(max(1, len(others))) as ex:
            res = dict(zip(others, ex.map(run, others)))
        if "gmix_strict" in res:
            same = res["gmix_strict"][1] == said
            for f in ("data/tmp", "analysis/training.tsv", "data/trained_checkpoint.long"):
                same = same and open(os.path.join(tmp, "gmix_strict", f), "rb").read() == open(os.path.join(tmp, exe, f), "rb").read()
            out["identical_to_stock"] = same
        if "gmix_fast" in res:
            out["cpu_baseline"] = {"value": bits / res["gmix_fast"][0], "unit": "bits/s", "cores": 1, "kind": "reference",
                                   "sample": f"the reference's own `gmix -t` on the same two files, -Ofast -march=x86-64-v3 (the "
                                             f"makefile's flags with a portable -march, not native), {res['gmix_fast'][0]:.1f} s"
                                             + (", run beside the strict build's" if len(others) > 1 else "")}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=30000)
    ap.add_argument("--variant", default="chain", choices=sorted(EXE))
    ap.add_argument("--chunk", type=int, default=2048)
    a = ap.parse_args()
    print(json.dumps(measure(a.streams, a.bytes, a.variant, a.chunk)))
