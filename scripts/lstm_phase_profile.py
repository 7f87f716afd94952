#!/usr/bin/env python3
"""Where a byte's time goes in gmx_lstm_kernel.  Needs a library built with -DGMX_LSTM_PROF:
  make -C gmix_amd/csrc prof     # -> gmix_amd/libgmxmix_prof.so
  GMX_LIB=$PWD/gmix_amd/libgmxmix_prof.so python scripts/lstm_phase_profile.py [streams=256] [bytes=200]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmix_amd
from gmix_amd import synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = gmix_amd.LstmGroup(S)
w = synth.lstm_initial_weights()
for s in range(S):
    g.set_weights(w, stream=s)
b = gmix_amd.LstmBatch(g, N)
ppm, data = synth.lstm_records(N, seed=1, mask=63)
for s in range(S):
    b.ppm[s] = ppm
    b.bytes[s] = data
b.upload(N)
ms = g.run(b, N, learn=True, timed=True)
out = (C.c_ulonglong * 16)()
g.L.gmx_lstm_prof_read(out)
names = ["loop top", "inputs", "gate chains", "norm, activations, cell", "output layer, max, expf",
         "softmax sum, divide, context, early SGD", "bit predictions", "backward: error + hidden-error chain",
         "backward: layer", "backward: layer-norm Adam, late SGD", "backward: gates, clips",
         "  deferred accumulation: error vectors to registers", "  deferred accumulation: tile staging",
         "  deferred accumulation: 100-term sums + Adam", "  Adam of the symbol columns"]
tot = float(sum(out))
print(f"kernel {ms:.2f} ms for {N} bytes x {S} streams = {ms * 1e3 / N:.1f} us per byte")
for k, nme in enumerate(names):
    print(f"  {nme:52s} {out[k] / tot * ms * 1e3 / N:7.2f} us per byte")
