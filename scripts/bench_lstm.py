#!/usr/bin/env python3
"""Auxiliary bench line for the LSTM byte model: S streams x N bytes
per launch, Predict x 8 bits + Learn (backward pass every 100th byte).
  python scripts/bench_lstm.py [--streams S --bytes N --steps K]"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--bytes", type=int, default=200)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample-bytes", type=int, default=20000)
    ap.add_argument("--no-learn", action="store_true", help="Predict only (generation mode)")
    args = ap.parse_args()
    print(json.dumps(measure(args.streams, args.bytes, args.steps, args.warmup, args.cpu_sample_bytes, not args.no_learn)))


def measure(streams=1024, nbytes=200, steps=4, warmup=1, cpu_sample_bytes=20000, learn=True):
    """One result object in bench.py's conventions (also what bench.py's `also.lstm` carries)."""
    import gmix_amd
    from gmix_amd import synth
    S, N = streams, nbytes
    g = gmix_amd.LstmGroup(S)
    w = synth.lstm_initial_weights()
    for s in range(S):
        g.set_weights(w, stream=s)
    b = gmix_amd.LstmBatch(g, N)
    ppm, data = synth.lstm_records(N, seed=1, mask=63)
    rng = np.random.default_rng(0)
    for s in range(S):   # same distributions, different byte streams
        b.ppm[s] = ppm
        b.bytes[s] = np.roll(data, int(rng.integers(0, N)))
    b.upload(N)
    for _ in range(warmup):
        g.run(b, N, learn=learn)
    g.sync()
    ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ms.append(g.run(b, N, learn=learn, timed=True))
    g.sync()
    el = time.perf_counter() - t0
    avg = sum(ms) / len(ms)
    build = g.L.gmx_build_info().decode()
    # algorithmic HBM bytes per stream-byte: gate weights read once per forward (3 x 50 x 308 x 4),
    # the output layer read twice (forward, SGD) and its next ring slot written (3 x 256 x 51 x 4),
    # the records and the stored layer input (2 x 307 x 4 + 1024); per backward epoch (one per byte
    # on average): the output layer of the epoch (256 x 51 x 4), the recurrent weights (3 x 50 x 50 x 4);
    # per pass / 100: Adam's m, v, w read and written (3 x 6 x 563 x 50 x 4) and the layer inputs (4 x 307 x 100 x 4)
    bpb = (3 * 50 * 308 * 4 + 3 * 256 * 51 * 4 + 2 * 307 * 4 + 1024 + 256 * 51 * 4 + 3 * 50 * 50 * 4
           + (3 * 6 * 563 * 50 * 4 + 4 * 307 * 100 * 4) // 100)
    out = {"metric": "LSTM byte-model bytes/sec (Predict x 8 bits + Learn, backward pass every 100th byte)",
           "value": S * N * steps / el, "unit": "bytes/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": el / steps * 1e3, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "LstmModel (Lstm(256,256,50,1,100,0.03,10))", "streams": S, "bytes_per_stream_per_step": N,
                      "bank_bytes_per_stream": g.bank_bytes, "bits_per_s": S * N * 8 * steps / el},
           "roofline": {"bound": "hbm", "achieved": bpb * S * N / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": bpb * S * N / (avg * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel": "gmx_lstm_kernel",
                        "kernel_ms_avg": avg, "kernel_ms_min": min(ms), "kernel_ms_median": sorted(ms)[len(ms) // 2],
                        "kernel_ms_max": max(ms), "algorithmic_bytes_per_byte": bpb, "build": build}}
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_lstm_harness")
    if os.path.exists(exe):
        n = cpu_sample_bytes
        t1 = time.perf_counter()
        subprocess.run([exe, "--bytes", str(n), "--seed", "1", "--mask", "63", "--out", "/tmp/_lstm_cpu.bin"], check=True,
                       stdout=subprocess.DEVNULL)
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": n / dt, "unit": "bytes/s", "cores": 1, "kind": "reference",
                               "sample": f"{n} bytes through the reference's own LstmModel (strict -O2 harness), 1 thread of {os.cpu_count()}"}
    b.close()
    g.close()
    return out


if __name__ == "__main__":
    main()
