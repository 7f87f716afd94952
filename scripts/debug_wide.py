import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gmix_amd as gpu
from gmix_amd import topology
from oracle import gmxo as oracle
from test_gpu_wide import run, beq
kw = dict(ctx_mode=0)
topo = topology.synth3(256, table0=1 << 12)
S, T = int(sys.argv[1]), int(sys.argv[2])
chunk = int(sys.argv[3])
streams = [oracle.synth(256, 33, T, seed=4321 + 13 * s, **kw) for s in range(S)]
g1, P1, O1 = run(gpu, topo, streams, chunk, force_general=False, mask=True)
for s in range(S):
    ob = oracle.Bank(256, topo.skip, topo.mixers)
    p_ref, o_ref = ob.run(*streams[s])
    a = O1[s].view(np.uint32); b = o_ref.view(np.uint32)
    bad = np.argwhere(a != b)
    print("stream", s, "mismatches", len(bad), "first", bad[:6].tolist())
    if len(bad):
        t, m = bad[0]
        print("  t", t, "m", m, O1[s][t, m], o_ref[t, m])
        ctx = streams[s][2]
        rows = ctx[:, m] % topo.mixers[m][1]
        prev = np.nonzero(rows[:t] == rows[t])[0]
        print("  row", rows[t], "earlier visits at", prev[-5:].tolist())
        print("  per-mixer first bad t:", {int(mm): int(bad[bad[:, 1] == mm][0, 0]) for mm in np.unique(bad[:, 1])})
