#!/usr/bin/env python3
"""One lock-step step of the whole device chain (gmx_chainstep: LSTM byte model, 41 Indirect models, 33 mixers of S
streams, one hipGraph per coded bit) driven from Python on synthetic records: microseconds per step by the host's
clock, and -- under `rocprofv3 --kernel-trace --memory-copy-trace` -- the device-side timeline of a step
(scripts/trace_chainstep.sh).  The records change every step (gate rows, table entries and bit contexts move as in a
real run: byte-structured bit contexts, bit-level mixer contexts redrawn every bit, the rest per byte).
  python scripts/bench_chainstep.py [--streams 64] [--steps 4000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--steps", type=int, default=4000)
    a = ap.parse_args()
    import gmix_amd
    from gmix_amd import topology
    S, K = a.streams, 41
    z = np.load(os.path.join(ROOT, "tests", "golden", "ind_stock41.npz"))
    models = topology.stock_indirect()
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    lg = gmix_amd.LstmGroup(S)
    ig = gmix_amd.IndirectGroup(models, z["ns_next"], z["rm_next"], S, slots=slots)
    mg = gmix_amd.MixerGroup(topology.stock(90), S)
    rng = np.random.default_rng(1)
    w0 = ((rng.random((3, 50, 563), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    for s in range(S):
        lg.set_weights(w0, stream=s)
    cs = gmix_amd.ChainStep(mg, ig, lg, lstm_slot=1, mixer_ctx_col=22, ind_ctx_col=16)
    cs.predictions[:, :90] = rng.standard_normal((S, 90)).astype(np.float32)
    cs.active_mask[:] = 0
    cs.active_mask[:, 0] = 0xfd  # the host-side models' slots (0, 2..7); the device-side models set their own
    ppm = rng.random((S, 256), dtype=np.float32)
    ppm /= ppm.sum(axis=1, keepdims=True)
    byte_ctx = rng.integers(0, 1 << 16, (S, 33), dtype=np.uint32)
    ind_ctx = rng.integers(0, 1 << 24, (S, K), dtype=np.uint32)
    bits = rng.integers(0, 2, (a.steps + 1, S), dtype=np.uint8)
    recent = np.ones(S, np.uint32)
    t_host = 0.0
    for t in range(a.steps + 1):
        if t % 8 == 0:
            byte_ctx = rng.integers(0, 1 << 16, (S, 33), dtype=np.uint32)
            ind_ctx = rng.integers(0, 1 << 24, (S, K), dtype=np.uint32)
            cs.ppm[:] = ppm
            recent[:] = 1
        ctx = byte_ctx.copy()
        ctx[:, (2, 11, 26, 29)] = (byte_ctx[:, (2, 11, 26, 29)] << 8) | recent[:, None]   # the bit-level gate contexts
        cs.contexts[:] = ctx
        cs.ind_contexts[:] = ind_ctx
        cs.bit_contexts[:] = recent - 1
        if t > 0:
            cs.bits[:] = bits[t - 1]
        cs.what[:] = (1 if t > 0 else 0) | (2 if t < a.steps else 0)
        t0 = time.perf_counter()
        cs.step()
        t_host += time.perf_counter() - t0
        recent = recent * 2 + bits[t]
    print(json.dumps({"streams": S, "steps": a.steps + 1, "us_per_step_gmx_chainstep_step": t_host * 1e6 / (a.steps + 1),
                      "stream_bits_per_s_at_that_rate": S * a.steps / t_host, "build": mg.L.gmx_build_info().decode()}))
    cs.close()


if __name__ == "__main__":
    main()
