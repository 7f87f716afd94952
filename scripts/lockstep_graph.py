#!/usr/bin/env python3
"""Lock-step stepping of S streams through the hipGraph surface
(gmx_lockstep_*): per step Predict for all streams, the S probabilities on the host, Learn for all
streams.  Reports microseconds per step; compare scripts/lockstep_decode.py (batched surface).
  python scripts/lockstep_graph.py [streams=256] [steps=400]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gmix_amd
from gmix_amd import topology

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
g = gmix_amd.MixerGroup(topology.stock(90), S)
ls = gmix_amd.Lockstep(g)
b = ls.batch
rng = np.random.default_rng(3)
pred = ((rng.integers(0, 2001, (S, 1, b.n_pad)) - 1000) / 250.0).astype(np.float32)
pred[:, :, 90:] = 0
b.predictions[:] = pred
b.active_mask[:] = 0xFFFFFFFF
b.contexts[:] = rng.integers(0, 1 << 16, (S, 1, 33)).astype(np.uint32)
bits = b.bits
for k in range(20):
    p = ls.predict(); bits[:, 0] = p > 0.5; ls.learn()
g.sync()
t0 = time.perf_counter()
for k in range(steps):
    if k % 8 == 0:
        b.contexts[:] = rng.integers(0, 1 << 16, (S, 1, 33)).astype(np.uint32)
    p = ls.predict()          # Predict for all streams, probabilities on the host
    bits[:, 0] = p > 0.5      # stands in for S arithmetic decoders
    ls.learn()                # Learn for all streams
g.sync()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": "stock 24/8/1, lock-step hipGraph surface: predict / host round trip / learn",
                  "streams": S, "us_per_step": dt * 1e6, "us_per_stream_bit": dt * 1e6 / S, "stream_bits_per_s": S / dt}))
