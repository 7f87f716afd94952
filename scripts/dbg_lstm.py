import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import gmix_amd as gpu
from oracle import gmxo as oracle
N, cut = 390, 230
ppm, data = oracle.lstm_synth(N, seed=9, mask=127)
m = oracle.LstmModel()
g = gpu.LstmGroup(3)
m.run(ppm[:cut], data[:cut])
g.import_(m.export_long(), m.export_short(), stream=1)
g.forward(ppm[0], int(data[-1]), stream=1)
m.predict_byte(ppm[0], int(data[-1]))
a = g.export(1); b = (m.export_long(), m.export_short())
for k in (0,1):
    x = np.frombuffer(a[k], np.uint8); y = np.frombuffer(b[k], np.uint8)
    d = np.nonzero(x != y)[0]
    print("part", k, len(x), len(y), "ndiff", len(d), d[:20], x[d[:12]], y[d[:12]])
