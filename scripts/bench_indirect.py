#!/usr/bin/env python3
"""Auxiliary bench line for the Indirect models: the 41 stock
models, S streams x T bits per launch, synthetic byte-structured records generated on the device
(oracle/gmx_ind_synth.h).  Not the headline metric (that is bench.py); same JSON conventions.
  python scripts/bench_indirect.py [--streams S --bits T --steps K --into-mixer]"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--bits", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--into-mixer", action="store_true", help="also write into a stock mixer batch and run the mixers")
    ap.add_argument("--cpu-sample-bits", type=int, default=400_000)
    args = ap.parse_args()
    print(json.dumps(measure(args.streams, args.bits, args.steps, args.warmup, args.cpu_sample_bits, args.into_mixer)))


def measure(streams=256, bits=4096, steps=8, warmup=2, cpu_sample_bits=400_000, into_mixer=False):
    """One result object in bench.py's conventions (also what bench.py's `also.indirect` carries)."""
    import gmix_amd
    import goldenlib
    from gmix_amd import topology
    _, z = goldenlib.load("ind_tiny_dense")  # carries the reference's two next-state tables
    models = topology.stock_indirect()
    K, S, T = len(models), streams, bits
    mods = (300, 0, 70000, 5)
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    g = gmix_amd.IndirectGroup(models, z["ns_next"], z["rm_next"], S, slots=slots)
    ring = [gmix_amd.IndirectBatch(g, T) for _ in range(2)]
    for i, b in enumerate(ring):
        b.fill_synthetic(T, seed=77 + i, restart=True, ctx_mod=mods)
    mg = mb = None
    if into_mixer:
        mg = gmix_amd.MixerGroup(topology.stock(90), S)
        mb = gmix_amd.Batch(mg, T, outputs=False, mask=True)
        mb.fill_synthetic(T, seed=5, restart=True, ctx_mode=2)
        mg.sync()
    g.sync()
    for k in range(warmup):
        g.run(ring[k % 2], T, learn=True, into=mb)
        if mg:
            mg.run(mb, T, learn=True)
    g.sync()
    if mg:
        mg.sync()
    ms = []
    t0 = time.perf_counter()
    for k in range(steps):
        ms.append(g.run(ring[k % 2], T, learn=True, into=mb, timed=True))
        if mg:
            mg.run(mb, T, learn=True)
    g.sync()
    if mg:
        mg.sync()
    el = time.perf_counter() - t0
    avg = sum(ms) / len(ms)
    build = g.L.gmx_build_info().decode()
    # algorithmic bytes per stream-bit: per model one 2-byte state pair read and written; the
    # record (contexts, bit_context, bit) read; two predictions + two flags per model written
    bpb = K * 4 + K * 4 + 5 + K * 10
    out = {"metric": "indirect-model bits/sec (41 stock Indirect models, Predict+Learn)", "value": S * T * steps / el,
           "unit": "bits/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
           "higher_is_better": True, "dtype": "u8/f32", "data": "synthetic",
           "config": {"workload": "41 stock Indirect models" + (" feeding the 33 stock mixers in HBM" if mg else ""),
                      "streams": S, "bits_per_stream_per_step": T, "bank_bytes_per_stream": g.bank_bytes},
           "roofline": {"bound": "hbm", "achieved": bpb * S * T / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": bpb * S * T / (avg * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel": "gmx_indirect_kernel",
                        "kernel_ms_avg": avg, "kernel_ms_min": min(ms), "kernel_ms_median": sorted(ms)[len(ms) // 2],
                        "kernel_ms_max": max(ms), "algorithmic_bytes_per_bit": bpb, "build": build}}
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_indirect_harness")
    if os.path.exists(exe):
        n = cpu_sample_bits
        t1 = time.perf_counter()
        subprocess.run([exe, "--models", ",".join(f"{t}:{lr!r}" for t, lr in models), "--bits", str(n),
                        "--ctx-mod", ",".join(map(str, mods)), "--out", "/tmp/_ind_cpu.bin"], check=True,
                       stdout=subprocess.DEVNULL)
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": n / dt, "unit": "bits/s", "cores": 1, "kind": "reference",
                               "sample": f"{n} bits through the reference's own Indirect class (strict -O2 harness, "
                                         f"incl. table construction and its checksum), 1 thread of {os.cpu_count()}"}
    for b in ring:
        b.close()
    if mb:
        mb.close()
    if mg:
        mg.close()
    g.close()
    return out


if __name__ == "__main__":
    main()
