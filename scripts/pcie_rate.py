#!/usr/bin/env python3
"""PCIe-inclusive rate of the batched surface for BASELINE configs[1]:
records start in pinned host memory, probabilities end there.  Never the bench's `value`.
  python scripts/pcie_rate.py [streams=1024] [bits=256] [steps=6]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gmix_amd
from gmix_amd import topology

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
topo = topology.single(256, 1 << 16, 0.005)
g = gmix_amd.MixerGroup(topo, S)
b = gmix_amd.Batch(g, T, outputs=False, mask=False)
rng = np.random.default_rng(1)
b.predictions[:, :T, :] = ((rng.integers(0, 2001, (S, T, 256)) - 1000) / 250.0).astype(np.float32)
b.contexts[:, :T, :] = rng.integers(0, 1 << 32, (S, T, 1), dtype=np.uint64).astype(np.uint32)
b.bits[:, :T] = rng.integers(0, 2, (S, T)).astype(np.uint8)
_ = b.p
rec_bytes = S * T * (256 * 4 + 4 + 1)
for _ in range(2):
    b.upload(T); g.run(b, T, learn=True); b.download(T); b.wait()
tu = tk = td = 0.0
for _ in range(steps):
    t0 = time.perf_counter(); b.upload(T); b.wait(); t1 = time.perf_counter()
    g.run(b, T, learn=True); g.sync(); t2 = time.perf_counter()
    b.download(T); b.wait(); t3 = time.perf_counter()
    tu += t1 - t0; tk += t2 - t1; td += t3 - t2
tot = tu + tk + td
print(json.dumps({"workload": "configs[1], records from pinned host memory, p back to host (serial: H2D, kernel, D2H)",
                  "streams": S, "bits_per_stream": T, "steps": steps,
                  "bits_per_s_pcie_inclusive": S * T * steps / tot,
                  "h2d_GB_per_s": rec_bytes * steps / tu / 1e9,
                  "ms_per_step": {"h2d": tu / steps * 1e3, "kernel": tk / steps * 1e3, "d2h": td / steps * 1e3}}))
