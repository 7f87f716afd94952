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

# how even the blocks of the last launch were, and the clock the chip ran at
import numpy as np
blk = (C.c_ulonglong * (4 * S))()
if hasattr(g.L, "gmx_stock_prof_blocks") and g.L.gmx_stock_prof_blocks(blk, S) == 0:
    raw = np.frombuffer(blk, dtype=np.uint64).reshape(S, 4)
    a = raw.astype(np.float64)
    cyc, ticks, start = a[:, 0], a[:, 1], a[:, 2]
    ghz = cyc / (ticks * 10.0)
    print(f"  blocks: shader cycles min/median/max {cyc.min():.0f} / {np.median(cyc):.0f} / {cyc.max():.0f}; "
          f"duration us min/median/max {ticks.min() / 100:.1f} / {np.median(ticks) / 100:.1f} / {ticks.max() / 100:.1f}; "
          f"clock GHz min/median/max {ghz.min():.3f} / {np.median(ghz):.3f} / {ghz.max():.3f}; "
          f"start spread {(start.max() - start.min()) / 100:.1f} us, first start to last end {((start + ticks).max() - start.min()) / 100:.1f} us")
    hw = raw[:, 3]
    hwid, xcc = (hw & np.uint64(0xffffffff)).astype(np.int64), ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
    for name, key in (("XCC", xcc), ("SE", se), ("SH", sh), ("CU", cu), ("SIMD", simd)):
        ks = sorted(set(key.tolist()))
        print(f"  duration us by {name}: " + "  ".join(f"{k}: {ticks[key == k].mean() / 100:.1f} ({int((key == k).sum())})" for k in ks))
    where = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10
    per_cu = {}
    for w, t_ in zip(where.tolist(), ticks.tolist()):
        per_cu.setdefault(w, []).append(t_ / 100)
    occ = np.bincount([len(v) for v in per_cu.values()])
    print(f"  CUs used {len(per_cu)}; blocks per CU histogram {dict((i, int(c)) for i, c in enumerate(occ) if c)}")
    by_n = {}
    for v in per_cu.values():
        by_n.setdefault(len(v), []).extend(v)
    print("  duration us by blocks on the CU: " + "  ".join(f"{n}: {np.mean(v):.1f}" for n, v in sorted(by_n.items())))
