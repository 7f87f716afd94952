"""Reader of oracle/ref_build/ref_trace.cpp's dump (the reference's whole Predictor, recording what
crosses the mixer boundary on every bit), vectorised so that hundreds of thousands of bits load in
a moment.  oracle/_ref/ref_trace is the reference itself, compiled in the build container; the
binary travels to the GPU box, so a test can make a trace of any file that exists on both boxes."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TRACE = os.path.join(ROOT, "oracle", "_ref", "ref_trace")


def make_trace(src, n_bytes, out, analysis=0):
    subprocess.run([REF_TRACE, src, str(n_bytes), out, str(analysis)], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, cwd=os.path.dirname(out), timeout=1100)
    return read_trace(out)


def read_trace(path):
    b = np.fromfile(path, np.uint8)
    hdr = b[:28].tobytes()
    magic, ver, n, M, L0, L1, nskip = struct.unpack("<7I", hdr)
    assert magic == 0x54584D47 and ver == 1
    off = 28
    skip = list(struct.unpack(f"<{nskip}I", b[off:off + 4 * nskip].tobytes()))
    off += 4 * nskip
    T, = struct.unpack("<Q", b[off:off + 8].tobytes())
    off += 8
    mixers, wsize = [], []
    for _ in range(M):
        layer, table, lr, ws = struct.unpack("<iIfi", b[off:off + 16].tobytes())
        off += 16
        mixers.append((layer, table, float(np.float32(lr))))
        wsize.append(ws)
    rec = np.dtype([("pred", "<f4", n), ("act", "u1", n), ("ctx", "<u4", M), ("bit", "u1"), ("outs", "<f4", M),
                    ("p", "<f4")])
    assert rec.itemsize == 4 * n + n + 4 * M + 1 + 4 * M + 4
    r = np.frombuffer(b, rec, T, off)
    off += T * rec.itemsize
    ns, = struct.unpack("<Q", b[off:off + 8].tobytes())
    off += 8
    short = b[off:off + ns].tobytes()
    off += ns
    nl, = struct.unpack("<Q", b[off:off + 8].tobytes())
    off += 8
    long_b = b[off:off + nl].tobytes()
    return dict(n=n, M=M, skip=skip, T=int(T), mixers=mixers, weight_sizes=wsize,
                pred=np.ascontiguousarray(r["pred"]), act=np.ascontiguousarray(r["act"]),
                ctx=np.ascontiguousarray(r["ctx"]), bits=np.ascontiguousarray(r["bit"]),
                outs=np.ascontiguousarray(r["outs"]), p=np.ascontiguousarray(r["p"]), short=short, long=long_b)
