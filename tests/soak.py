#!/usr/bin/env python3
"""Long parity run (not part of the test suite: minutes of CPU oracle time): S streams x T bits through
the batched surface in chunks, every output and the final state against the oracle.
  python tests/soak.py [--shape wide|stock|stock-pairs|single] [--streams 3] [--bits 300000] [--chunk 7000] [--staged 1]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gmix_amd as gpu
from gmix_amd import topology
from oracle import gmxo as oracle

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="wide")
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--bits", type=int, default=300000)
ap.add_argument("--chunk", type=int, default=7000)
ap.add_argument("--staged", type=int, default=-1, help="stock shape: rows through the LDS images (1), lane-private (0), by stream count (-1)")
ap.add_argument("--plain", type=int, default=0, help="1: only the probabilities are stored (the stock kernel's plain build); the state is still compared")
ap.add_argument("--mask", type=int, default=-1, help="active masks on (1) / off (0); default: on except for the one-mixer shape")
a = ap.parse_args()
topo = {"wide": lambda: topology.synth3(256, table0=1 << 8), "stock": lambda: topology.stock(90),
        "stock-pairs": lambda: topology.stock(90), "single": lambda: topology.single(256, 1 << 6, 0.005)}[a.shape]()
n, m, S, T = topo.n_inputs, topo.n_mixers, a.streams, a.bits
g = gpu.MixerGroup(topo, S)
if a.shape == "stock-pairs":
    g.L.gmx_debug_stock_pairs.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_stock_pairs(g.h, 1) == 0
if a.staged >= 0:
    g.L.gmx_debug_stock_staged.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_stock_staged(g.h, a.staged) == 0
mask = a.shape != "single" if a.mask < 0 else bool(a.mask)
b = gpu.Batch(g, a.chunk, outputs=not a.plain, mask=mask)
# small context ranges: rows are revisited thousands of times (weight shrink every 1024th visit), come
# back after having been written back, stay for several bits
kw = [dict(ctx_mode=3, ctx_mod=6, zero_mod=9, bit_mode=1), dict(ctx_mode=1, ctx_mod=3, bit_mode=1), dict(ctx_mode=2, ctx_mod=40, zero_mod=5, bit_mode=1),
      dict(ctx_mode=5, ctx_mod=50, zero_mod=7, bit_mode=1)]  # the last: a real run's pattern, 4 rows moving every bit
gens = [oracle.Stream(n, m, seed=1000 + s, **kw[s % 4]) for s in range(S)]
if a.shape == "single":
    gens = [oracle.Stream(n, m, seed=1000 + s, ctx_mode=1, ctx_mod=5 + s, bit_mode=1) for s in range(S)]
banks = [oracle.Bank(n, topo.skip, topo.mixers) for _ in range(S)]
t0 = time.time()
done = 0
while done < T:
    c = min(a.chunk, T - done)
    refs = []
    for s in range(S):
        rec = gens[s].next(c)
        if not mask:
            rec = (rec[0], None, rec[2], rec[3])
        b.set_records(s, *rec)
        refs.append(banks[s].run(rec[0], rec[1] if mask else np.ones((c, n), np.uint8), rec[2], rec[3]))
    b.upload(c)
    g.run(b, c, learn=True)
    b.download(c)
    b.wait()
    for s in range(S):
        p_ref, o_ref = refs[s]
        if not a.plain:
            assert np.array_equal(b.outputs[s, :c].view(np.uint32), o_ref.view(np.uint32)), (s, done)
        assert np.array_equal(b.p[s, :c].view(np.uint32), p_ref.view(np.uint32)), (s, done)
    done += c
    if (done // a.chunk) % 10 == 0:
        print(f"{done} bits ok ({time.time() - t0:.0f} s)", flush=True)
for s in range(S):
    assert g.export(s) == (banks[s].export_long(), banks[s].export_short()), s
print(f"soak ok: {a.shape}{' (plain build)' if a.plain else ''}{'' if mask else ', no masks'}, {S} streams x {T} bits, "
      f"{'probabilities' if a.plain else 'outputs'} and state == oracle")
