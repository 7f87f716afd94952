#!/usr/bin/env python3
"""PCIe-inclusive rate of the batched surface with double-buffered record batches (BASELINE
configs[3]: "pinned hipMemcpyAsync double-buffered host->device probability batches"): records
start in pinned host memory, probabilities end there; while the kernel runs on one batch the other
one's records cross PCIe.  Never the bench's `value`.
  python scripts/pcie_pipeline.py [--config single|synth3|stock] [--streams S] [--bits T] [--steps K]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gmix_amd
from gmix_amd import topology

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="single")
ap.add_argument("--streams", type=int, default=1024)
ap.add_argument("--bits", type=int, default=256)
ap.add_argument("--steps", type=int, default=12)
args = ap.parse_args()
topo = {"single": lambda: topology.single(256, 1 << 16, 0.005), "synth3": lambda: topology.synth3(256, table0=1 << 12),
        "stock": lambda: topology.stock(90)}[args.config]()
S, T, n, m = args.streams, args.bits, topo.n_inputs, topo.n_mixers
mask = args.config != "single"
g = gmix_amd.MixerGroup(topo, S)
bs = [gmix_amd.Batch(g, T, outputs=False, mask=mask) for _ in range(2)]
rng = np.random.default_rng(1)
for b in bs:
    b.predictions[:, :T, :n] = ((rng.integers(0, 2001, (S, T, n)) - 1000) / 250.0).astype(np.float32)
    b.contexts[:, :T, :] = rng.integers(0, 1 << 32, (S, T, m), dtype=np.uint64).astype(np.uint32)
    b.bits[:, :T] = rng.integers(0, 2, (S, T)).astype(np.uint8)
    if mask:
        b.active_mask[:] = 0xFFFFFFFF
    _ = b.p
rec_bytes = S * T * (bs[0].predictions.shape[2] * 4 + m * 4 + 1 + (bs[0].active_mask.shape[2] * 4 if mask else 0))


def serial(steps):
    b = bs[0]
    t0 = time.perf_counter()
    for _ in range(steps):
        b.upload(T); g.run(b, T, learn=True); b.download(T); b.wait()
    return time.perf_counter() - t0


def pipelined(steps):
    t0 = time.perf_counter()
    bs[0].upload(T)
    for k in range(steps):
        cur, nxt = bs[k & 1], bs[(k + 1) & 1]
        g.run(cur, T, learn=True)
        if k + 1 < steps:
            nxt.wait()          # its previous download has reached the host: the host may refill it (not simulated)
            nxt.upload(T)       # crosses PCIe while `cur` is being computed
        cur.download(T)
    bs[0].wait(); bs[1].wait()
    return time.perf_counter() - t0


serial(2); pipelined(2)
ts, tp = serial(args.steps), pipelined(args.steps)
print(json.dumps({"workload": f"{args.config}: records from pinned host memory, p back to host", "streams": S,
                  "bits_per_stream": T, "steps": args.steps, "record_bytes_per_step": rec_bytes,
                  "serial_bits_per_s": S * T * args.steps / ts, "pipelined_bits_per_s": S * T * args.steps / tp,
                  "serial_ms_per_step": ts / args.steps * 1e3, "pipelined_ms_per_step": tp / args.steps * 1e3,
                  "h2d_GBps_pipelined": rec_bytes * args.steps / tp / 1e9}))
