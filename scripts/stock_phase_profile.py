#!/usr/bin/env python3
"""Where a bit's time goes in gmx_stock_kernel.  Needs a library built with -DGMX_STK_PROF:
  make -C gmix_amd/csrc prof     # -> gmix_amd/libgmxmix_prof.so
  GMX_LIB=$PWD/gmix_amd/libgmxmix_prof.so python scripts/stock_phase_profile.py [streams=256] [bits=256] [ctx_mode=2]
s_memtime ticks of wave 0 between stamps (the stamps add about 10 %)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmix_amd
from gmix_amd import topology

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = gmix_amd.MixerGroup(topology.stock(90), S)
b = gmix_amd.Batch(g, T, outputs=False, mask=False)
b.fill_synthetic(T, seed=5, restart=True, ctx_mode=mode)
g.run(b, T, learn=True)
g.sync()
out = (C.c_ulonglong * 16)()
g.L.gmx_stock_prof_read(out, 1)
reps = 4
ms = 0.0
for _ in range(reps):
    b.fill_synthetic(T, seed=5, restart=False, ctx_mode=mode)
    ms += g.run(b, T, learn=True, timed=True)
g.L.gmx_stock_prof_read(out, 0)
names = ["prefetch issue (loads of the next bit)", "mask + skip input", "forward stream (+ non-finite check)",
         "logistic + result stores", "learn scalars (fp64 decay)", "update stream", "commit: wait for the prefetch",
         "commit: evict / adopt / bookkeeping + loop"]
tot = float(sum(out))
nbits = reps * T
print(f"streams {S}, {T} bits/launch, ctx_mode {mode}: kernel {ms / reps:.3f} ms = {ms * 1e3 / nbits:.2f} us per bit per stream; "
      f"{tot / nbits:.0f} shader cycles per bit (100 MHz ticks x clock ratio not applied: s_memtime counts shader cycles)")
for k, nme in enumerate(names):
    print(f"  {nme:52s} {out[k] / nbits:8.0f} cycles  {100.0 * out[k] / tot:5.1f} %")
