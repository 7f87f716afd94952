#!/usr/bin/env python3
"""BASELINE.json configs[2]'s shape AT LENGTH on substitute text (enwik8 is on neither box; scripts/corpus.py): one
10^7-byte stream through `gmix_chain_batched -c` and 64 files of 10^6 bytes through `gmix_chain_many`, every output
compared -- by md5 -- with what the reference's strict stock build (`gmix_strict -c`) made of the same bytes in the build
container (tests/golden/long_expected.json, written by scripts/make_long_expected.sh; 35 minutes of 8 cores there).
Times are COLD: whole processes, exec to exit, against the reference's own CLI (-Ofast) run the same way on this
host -- 16 processes at once over the first 16 files, and one process on the first 10^6 bytes of the long stream (a
bounded sample of it: the whole of it would take the reference a quarter of an hour).
  python scripts/e2e_long.py [many] [big] [--files 64] [--out gpurun_out/e2e_long.json]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import corpus  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
DROPIN = os.path.join(ROOT, "dropin", "_build")


def md5(b):
    return hashlib.md5(b).hexdigest()


def host():
    cpu = "unknown CPU"
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(p)
    except Exception:
        quota = None
    return cpu, quota


def run_fast(files, tmp, tag):
    """the reference's CLI (-Ofast) on each file, all at once; wall seconds"""
    fast = os.path.join(REF, "gmix_fast")

    def one(k):
        d = os.path.join(tmp, f"{tag}{k}")
        os.mkdir(d)
        subprocess.run([fast, "-c", files[k], os.path.join(d, "c")], cwd=d, capture_output=True, timeout=3000, check=True)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(files)) as ex:
        list(ex.map(one, range(len(files))))
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["many", "big"])
    ap.add_argument("--files", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "e2e_long.json"))
    a = ap.parse_args()
    exp = json.load(open(os.path.join(ROOT, "tests", "golden", "long_expected.json")))
    data = corpus.pylib_bytes()
    cpu, quota = host()
    res = {"corpus": "python3.10 standard library sources (scripts/corpus.py)", "corpus_md5": md5(data),
           "corpus_is_the_expected_one": md5(data) == exp["corpus_md5"], "host_cpu": cpu, "cpu_quota_cores": quota}
    if not res["corpus_is_the_expected_one"] and not os.environ.get("GMX_CORPUS"):
        print(json.dumps(res))
        raise SystemExit("this box's /usr/lib/python3.10 differs from the build container's: no expected outputs to compare with")
    with tempfile.TemporaryDirectory() as tmp:
        if "many" in a.what:
            S = a.files
            files = []
            for k in range(S):
                e = exp["files"][k]
                w = data[e["offset"]:e["offset"] + e["bytes"]]
                assert md5(w) == e["in_md5"], k
                f = os.path.join(tmp, f"f{k:03d}")
                open(f, "wb").write(w)
                files.append(f)
            t0 = time.perf_counter()
            r = subprocess.run([os.path.join(DROPIN, "gmix_chain_many"), os.path.join(tmp, "out")] + files,
                               capture_output=True, text=True, timeout=3000)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                raise SystemExit("gmix_chain_many failed: " + r.stderr[-800:])
            st = json.loads(r.stdout.strip().splitlines()[-1])
            same = sum(md5(open(os.path.join(tmp, "out", f"{k}.gmix"), "rb").read()) == exp["files"][k]["out_md5"]
                       for k in range(S))
            bits = 8.0 * sum(exp["files"][k]["bytes"] for k in range(S))
            n = min(S, 16)
            dt_cpu = run_fast(files[:n], tmp, "c")
            cpu_rate = 8.0 * sum(exp["files"][k]["bytes"] for k in range(n)) / dt_cpu
            res["many"] = {"files": S, "bytes_per_file": exp["files"][0]["bytes"], "bits": bits,
                           "process_seconds": dt, "bits_per_s_cold": bits / dt,
                           "coding_loops_seconds": st["wall_seconds"], "bits_per_s_coding_loops": st["bits_per_second"],
                           "setup_seconds": st["build_seconds"], "teardown_seconds": st["teardown_seconds"],
                           "predictors_built_side_by_side": st["parallel_construction"],
                           "device_wait_frac": st["wait_seconds"] / st["wall_seconds"], "launches": st["launches"],
                           "identical_to_gmix_strict": same, "compared": S, "compressed_bytes": st["output_bytes"],
                           "cpu": {"what": f"{n} processes of the reference's `gmix -c` (-Ofast -march=x86-64-v3) at once, one "
                                           f"of the same files each, whole processes", "seconds": dt_cpu,
                                   "bits_per_s": cpu_rate},
                           "vs_cpu_cold": bits / dt / cpu_rate}
            print(json.dumps(res["many"]), flush=True)
        if "big" in a.what:
            e = exp["big"]
            w = data[:e["bytes"]]
            assert md5(w) == e["in_md5"]
            f = os.path.join(tmp, "big")
            open(f, "wb").write(w)
            d = os.path.join(tmp, "bigrun")
            os.mkdir(d)
            t0 = time.perf_counter()
            r = subprocess.run([os.path.join(DROPIN, "gmix_chain_batched"), "-c", f, os.path.join(d, "c")], cwd=d,
                               capture_output=True, text=True, timeout=3000)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                raise SystemExit("gmix_chain_batched failed: " + r.stderr[-800:])
            out = open(os.path.join(d, "c"), "rb").read()
            sample = os.path.join(tmp, "sample")
            open(sample, "wb").write(w[:1000000])
            dt_cpu = run_fast([sample], tmp, "s")
            res["big"] = {"bytes": e["bytes"], "process_seconds": dt, "bits_per_s_cold": 8.0 * e["bytes"] / dt,
                          "us_per_bit": dt * 1e6 / (8.0 * e["bytes"]), "compressed_bytes": len(out),
                          "identical_to_gmix_strict": md5(out) == e["out_md5"] and len(out) == e["out_bytes"],
                          "analysis_rows": sum(1 for _ in open(os.path.join(d, "analysis", "entropy.tsv"))),
                          "cpu": {"what": "the reference's `gmix -c` (-Ofast -march=x86-64-v3), one process, on the FIRST 10^6 "
                                          "bytes of the same stream (a bounded sample)", "seconds": dt_cpu,
                                  "bits_per_s": 8e6 / dt_cpu},
                          "vs_cpu_cold": (8.0 * e["bytes"] / dt) / (8e6 / dt_cpu)}
            print(json.dumps(res["big"]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
