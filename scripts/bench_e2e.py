#!/usr/bin/env python3
"""End-to-end bench lines: whole files through the reference's own feature models and coder on the host cores with
the run-ahead compressor (gmix_amd/host/gmx_batched.h) taking the device-side models in batches -- BASELINE.json
configs[2] / [3] in small, on text that is on both boxes (GMX_CORPUS=/path/to/enwik8 for the named data).

  variant "mixers": the 33 mixers on the MI355X (dropin/_build/gmix_many); the host keeps all 88 feature models
  variant "chain":  LSTM + 41 Indirect models + 33 mixers on the MI355X (gmix_chain_many); the host keeps PPMd, the
                    match models, the context hashes and the coder

S files of n_bytes each are compressed side by side (one Predictor and host thread per file, ONE device group).
`value` is COLD and like for like with the reference's CLI: bits of all files / wall time of the WHOLE PROCESS, from
exec to exit -- runtime start, Predictor construction, device banks, coding, teardown -- as `gmix -c` pays for its own
Predictor (runner-utils.cpp:88-121).  `value_coding_loops` = the same bits / the coding loops alone (every Predictor
and bank standing), `setup_seconds` what came before them.  Every output is compared with the stock strict build's
`gmix -c` of the same file (`identical_to_stock`).  cpu_baseline = the reference's own CLI with the makefile's -Ofast
on the same file(s), whole processes too, as many at once as there are files (at most 16): what the same host does
without the device; `vs_cpu` = value / cpu_baseline.value.
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
REF = os.path.join(ROOT, "oracle", "_ref")          # the checker: stock builds (cpu_baseline, identical_to_stock)
DROPIN = os.path.join(ROOT, "dropin", "_build")     # the product inside the reference (dropin/Makefile)
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


def files_that_fit(want, world=1):
    """How many Predictors (a host thread, ~0.4 GB of host memory and ~1.1 us of one core per coded bit each) this
    process may run when `world` ranks share the host: the cgroup's CPU quota and memory limit divided by the ranks
    (8 ranks x 64 Predictors would be 512 threads and 512 PPMd arenas on one host)."""
    if world <= 1:
        return want
    cores = cpu_quota() or os.cpu_count() or 16
    n = int(4 * cores / world)
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            n = min(n, int(int(lim) * 0.5 / world / 0.5e9))
    except Exception:
        pass
    return max(4, min(want, n))


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


def measure(streams=1, n_bytes=30000, variant="chain", chunk=2048, verify=True, cpu=True, device=None, before=None):
    """`before`: called once the files are written and right before the process starts (across ranks: a barrier, so that
    every rank's clock starts together)."""
    exe = os.path.join(DROPIN, EXE[variant])
    if not os.path.exists(exe):
        raise RuntimeError(f"{exe} missing (make -C dropin, needs /root/reference)")
    S = streams
    with tempfile.TemporaryDirectory() as tmp:
        files, src = [], None
        for k in range(S):
            data, src = corpus(n_bytes, 1531 * k)
            f = os.path.join(tmp, f"f{k}")
            open(f, "wb").write(data)
            files.append(f)
        if before:
            before()
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-T", str(chunk)] + (["--device", str(device)] if device is not None else []) +
                           [os.path.join(tmp, "out")] + files, capture_output=True, text=True, timeout=1500)
        process_seconds = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError(f"{EXE[variant]} failed: {r.stderr[-500:]}")
        st = json.loads(r.stdout.strip().splitlines()[-1])
        bits = 8 * st["input_bytes"]
        quota = cpu_quota()
        out = {"metric": "whole-compressor bits/sec (reference feature models + coder on the host, "
                         + ("33 mixers" if variant == "mixers" else "LSTM + 41 Indirect models + 33 mixers")
                         + " on the MI355X in run-ahead batches)",
               "value": bits / process_seconds, "value_coding_loops": st["bits_per_second"], "unit": "bits/s", "n_gpus": 1,
               "higher_is_better": True,
               "dtype": "f32", "data": "real text",
               "config": {"workload": f"{S} file(s) x {n_bytes} bytes of {src}, compressed side by side, "
                                      f"{chunk}-bit chunks through a ring of four batches", "streams": S,
                          "bytes_per_file": n_bytes, "chunk_bits": chunk, "variant": variant,
                          "host_threads": S, "host_cpu": host_cpu(), "cpus_visible": os.cpu_count(),
                          "cpu_quota_cores": quota, "threads_pinned_to_gpu_numa_node": st["pinned_threads"]},
               "seconds": process_seconds, "coding_loops_seconds": st["wall_seconds"],
               "us_per_bit_per_stream": st["wall_seconds"] * 1e6 / (8 * n_bytes),
               # inside the process: everything before the coding loops (Predictors, device banks, pinned rings), the
               # first Predictor (always built alone), teardown; the rest of `seconds` is exec, runtime start and exit
               "setup_seconds": st["build_seconds"], "first_predictor_seconds": st["first_predictor_seconds"],
               "teardown_seconds": st["teardown_seconds"], "in_process_seconds": st["total_seconds"],
               "predictors_built_side_by_side": st["parallel_construction"], "launches": st["launches"],
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
            out["vs_cpu"] = out["value"] / (8 * n_bytes * n / dt)
            out["cpu_baseline"] = {"value": 8 * n_bytes * n / dt, "unit": "bits/s", "cores": n, "kind": "reference",
                                   "sample": f"{n} process(es) of the reference's own `gmix -c` at once (whole CLI: "
                                             f"Predictor construction included), -Ofast -march=x86-64-v3 (the makefile's "
                                             f"flags with a portable -march, not native), {n_bytes} bytes each, on "
                                             f"{host_cpu()}" + (f", cpu quota {quota:g} cores" if quota else "")}
    return out


def measure_decode(streams=64, n_bytes=4000, variant="chain", verify=True, cpu=True, device=None):
    """S files restored side by side (gmx::BatchedDecompressFiles: the reference's own Decoder per file, coder/decoder.cpp:
    19-39, all files' device-side models ONE gmx_chainstep step per coded bit).  `value` is cold like measure()'s: bits
    of all files / wall time of the whole `gmix_chain_many -d` process.  The files are first compressed by the same
    binary (untimed; their bytes are the strict stock build's, measure() checks that); identical_to_stock = every
    restored file equals its original.  cpu_baseline = the reference's own `gmix -d` at the makefile's -Ofast on the same
    texts (each compressed by that build's own `gmix -c` first, untimed: -Ofast codes other bytes), whole processes, as
    many at once as there are files (at most 16)."""
    exe = os.path.join(DROPIN, EXE[variant])
    if not os.path.exists(exe):
        raise RuntimeError(f"{exe} missing (make -C dropin, needs /root/reference)")
    S = streams
    dev = ["--device", str(device)] if device is not None else []
    with tempfile.TemporaryDirectory() as tmp:
        files, src = [], None
        for k in range(S):
            data, src = corpus(n_bytes, 1531 * k)
            f = os.path.join(tmp, f"f{k:04d}")
            open(f, "wb").write(data)
            files.append(f)
        r = subprocess.run([exe] + dev + [os.path.join(tmp, "c")] + files, capture_output=True, text=True, timeout=1500)
        if r.returncode != 0:
            raise RuntimeError(f"{EXE[variant]} failed: {r.stderr[-500:]}")
        coded = [os.path.join(tmp, "c", f"{k}.gmix") for k in range(S)]
        # The compression that made the files has just given back S x 1 GB of device memory, and the driver clears it in
        # the background before it hands any of it out again -- 7 s for 256 streams' 247 GB, during which a new process's
        # allocations wait (scripts/exp_vram_reuse.sh: banks up in 0.19 s after a pause, 2.3-4.0 s without).  That is the
        # preparation's wake, not the decompressor's cost: let it pass before the clock starts.
        settle = 0.03 * S
        time.sleep(settle)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-d"] + dev + [os.path.join(tmp, "back")] + coded, capture_output=True, text=True,
                           timeout=1500)
        process_seconds = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError(f"{EXE[variant]} -d failed: {r.stderr[-500:]}")
        st = json.loads(r.stdout.strip().splitlines()[-1])
        bits = 8.0 * S * n_bytes
        quota = cpu_quota()
        out = {"metric": "whole-decompressor bits/sec (the reference's Decoder, feature models and Predictor per file on the "
                         "host, " + ("33 mixers" if variant == "mixers" else "LSTM + 41 Indirect models + 33 mixers")
                         + " of all files one device step per coded bit)",
               "value": bits / process_seconds, "value_coding_loops": st["bits_per_second"], "unit": "bits/s", "n_gpus": 1,
               "higher_is_better": True, "dtype": "f32", "data": "real text",
               "config": {"workload": f"{S} file(s) x {n_bytes} bytes of {src}, restored side by side in lock step",
                          "streams": S, "bytes_per_file": n_bytes, "variant": variant, "worker_threads": st["pinned_cpus"],
                          "host_cpu": host_cpu(), "cpu_quota_cores": quota},
               "seconds": process_seconds, "coding_loops_seconds": st["wall_seconds"], "setup_seconds": st["build_seconds"],
               "in_process_seconds": st["total_seconds"], "steps": st["launches"], "settled_seconds_before_the_clock": settle,
               "us_per_step": st["wall_seconds"] * 1e6 / max(1, st["launches"])}
        if verify:
            out["identical_to_stock"] = all(open(os.path.join(tmp, "back", f"{k}.out"), "rb").read() == open(files[k], "rb").read()
                                            for k in range(S))
            out["files_compared"] = S
        fast = os.path.join(REF, "gmix_fast")
        if cpu and os.path.exists(fast):
            n = min(S, 16)

            def each(mode, k):
                d = os.path.join(tmp, f"x{k}")
                os.makedirs(d, exist_ok=True)
                a, b = (files[k], os.path.join(d, "c")) if mode == "-c" else (os.path.join(d, "c"), os.path.join(d, "d"))
                subprocess.run([fast, mode, a, b], cwd=d, capture_output=True, timeout=1500, check=True)
            with ThreadPoolExecutor(n) as ex:
                list(ex.map(lambda k: each("-c", k), range(n)))
                t0 = time.perf_counter()
                list(ex.map(lambda k: each("-d", k), range(n)))
                dt = time.perf_counter() - t0
            ok = all(open(os.path.join(tmp, f"x{k}", "d"), "rb").read() == open(files[k], "rb").read() for k in range(n))
            out["vs_cpu"] = out["value"] / (8 * n_bytes * n / dt)
            out["cpu_baseline"] = {"value": 8 * n_bytes * n / dt, "unit": "bits/s", "cores": n, "kind": "reference",
                                   "restores_its_input": ok,
                                   "sample": f"{n} process(es) of the reference's own `gmix -d` at once (whole CLI: Predictor "
                                             f"construction included), -Ofast -march=x86-64-v3, {n_bytes} bytes each, on "
                                             f"{host_cpu()}" + (f", cpu quota {quota:g} cores" if quota else "")}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=30000)
    ap.add_argument("--variant", default="chain", choices=sorted(EXE))
    ap.add_argument("--chunk", type=int, default=2048)
    ap.add_argument("--decode", action="store_true", help="measure_decode: S files restored in lock step")
    a = ap.parse_args()
    if a.decode:
        print(json.dumps(measure_decode(a.streams, a.bytes, a.variant)))
    else:
        print(json.dumps(measure(a.streams, a.bytes, a.variant, a.chunk)))
