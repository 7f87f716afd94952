#!/usr/bin/env python3
"""Decode-style stepping for many streams in lock step: per bit one launch
predicts for all S streams, the host reads the S probabilities (and would decode S bits), one
launch learns them.  Reports microseconds per step and per stream-bit.
  python scripts/lockstep_decode.py [streams=256] [steps=200]"""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = gmix_amd.MixerGroup(topology.stock(90), S)
b = gmix_amd.Batch(g, 1, outputs=False, mask=True)
rng = np.random.default_rng(3)
pred = ((rng.integers(0, 2001, (S, 1, b.n_pad)) - 1000) / 250.0).astype(np.float32)
pred[:, :, 90:] = 0
b.predictions[:] = pred
b.active_mask[:] = 0xFFFFFFFF
ctx = rng.integers(0, 1 << 16, (S, 1, 33)).astype(np.uint32)
b.contexts[:] = ctx
_ = b.p, b.bits
for k in range(20):
    b.upload(1); g.run(b, 1, learn=False); b.download(1); b.wait()
    b.bits[:, 0] = (b.p[:, 0] > 0.5); b.upload(1); g.run(b, 1, learn=True)
g.sync()
t0 = time.perf_counter()
for k in range(steps):
    if k % 8 == 0:
        b.contexts[:] = rng.integers(0, 1 << 16, (S, 1, 33)).astype(np.uint32)
    b.upload(1)
    g.run(b, 1, learn=False)       # Predict for all streams
    b.download(1)
    b.wait()
    b.bits[:, 0] = (b.p[:, 0] > 0.5)  # stands in for S arithmetic decoders
    b.upload(1)
    g.run(b, 1, learn=True)        # Predict again (same floats) + Learn
g.sync()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": "stock 24/8/1, 1-bit batched launches, predict / host round trip / predict+learn",
                  "streams": S, "us_per_step": dt * 1e6, "us_per_stream_bit": dt * 1e6 / S,
                  "stream_bits_per_s": S / dt}))
