#!/usr/bin/env python3
"""What a launch over streams at MANY different bit counts costs with the decay tables made by the
host's libm loop and by the device (gmx_decay_kernel).
  python scripts/bench_decay_tables.py [streams=1024] [bits=512] [steps=8]"""
import ctypes as C
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gmix_amd
from gmix_amd import topology

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
g = gmix_amd.MixerGroup(topology.single(256, 1 << 12, 0.005), S)
g.L.gmx_debug_decay_on_host.argtypes = [C.c_void_p, C.c_int]
for s in range(S):   # an empty bank that has already learned 1000 * s bits (mixer.cpp:178-188, long-term-memory.cpp:35-55)
    g.import_(struct.pack("<II", 0, 0), struct.pack("<QQQ", 1000 * s, 1, 0), stream=s)
b = gmix_amd.Batch(g, T, outputs=False, mask=False)
b.fill_synthetic(T, seed=3, restart=True)
g.sync()
out = {"streams": S, "bits_per_stream": T, "launches": K}
for name, on_host in (("host_libm_loop", 1), ("device_table", 0)):
    assert g.L.gmx_debug_decay_on_host(g.h, on_host) == 0
    g.run(b, T, learn=True)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        g.run(b, T, learn=True)
    g.sync()
    out[name + "_ms_per_launch"] = (time.perf_counter() - t0) / K * 1e3
print(json.dumps(out))
