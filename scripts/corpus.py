#!/usr/bin/env python3
"""Substitute text for the named corpora (enwik8 / enwik9 are on neither box): the Python standard library's own
sources -- /usr/lib/python3.10/**/*.py in sorted order, site-/dist-packages left out -- concatenated: 11 MB of
mixed prose and code that is part of the image, i.e. identical in the build container and on the GPU box (the md5
below is checked wherever it is used).  GMX_CORPUS=/path/to/enwik8 takes the named data instead.

  python scripts/corpus.py <out file> [n_bytes [offset]]     writes a window of it
The long end-to-end runs (scripts/e2e_long.sh, tests/test_gpu_long.py) compress one 10^7-byte stream and 64 windows of
10^6 bytes starting 157 000 bytes apart; what the reference's strict build makes of them is computed in the build
container (scripts/make_long_expected.sh) and travels as md5 sums in tests/golden/long_expected.json."""
import hashlib
import os
import sys

PYLIB = "/usr/lib/python3.10"
PYLIB_MD5 = None  # filled in by tests/golden/long_expected.json's "corpus_md5"
STRIDE = 157000


def pylib_bytes():
    names = []
    for d, dirs, files in os.walk(PYLIB):
        dirs[:] = sorted(x for x in dirs if x not in ("site-packages", "dist-packages"))
        names += [os.path.join(d, f) for f in files if f.endswith(".py")]
    names.sort()
    return b"".join(open(n, "rb").read() for n in names)


def window(n_bytes, offset=0):
    """(bytes, description) -- GMX_CORPUS if set and long enough, else the standard library's sources."""
    path = os.environ.get("GMX_CORPUS")
    if path:
        with open(path, "rb") as f:
            f.seek(offset)
            data = f.read(n_bytes)
        if len(data) == n_bytes:
            return data, path
    data = pylib_bytes()
    while len(data) < offset + n_bytes:
        data += data
    return data[offset:offset + n_bytes], "python3.10 standard library sources (scripts/corpus.py)"


if __name__ == "__main__":
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else None
    off = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    data = pylib_bytes()
    if n is None:
        n = len(data) - off
    w, _ = window(n, off)
    open(out, "wb").write(w)
    print(len(w), hashlib.md5(w).hexdigest(), "whole corpus:", len(data), hashlib.md5(data).hexdigest())
