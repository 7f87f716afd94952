"""Shared by tests/test_batched_cpu.py and tests/test_gpu_batched.py: the run-ahead compressor
(gmix_amd/host/gmx_batched.h inside builds of the reference, dropin/Makefile) beside
the stock build of the reference on the same bytes."""
import json
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

from dropin_common import ROOT, exe


def corpus(n_bytes, offset=0):
    """n_bytes of text that is on both boxes.  GMX_CORPUS=/path/to/enwik8 makes it the named data (BASELINE.json
    configs[0]: its first 10^6 bytes); by default the repository's own documents, repeated as needed."""
    path = os.environ.get("GMX_CORPUS")
    if path:
        with open(path, "rb") as f:
            f.seek(offset)
            data = f.read(n_bytes)
        assert len(data) == n_bytes, f"{path} is shorter than {offset + n_bytes} bytes"
        return data
    data = b"".join(open(os.path.join(ROOT, f), "rb").read() for f in ("DESIGN.md", "SURVEY.md", "INTEGRATION.md"))
    while len(data) < offset + n_bytes:
        data += data
    return data[offset:offset + n_bytes]


def need(*names):
    for name in names:
        assert os.path.exists(exe(name)), f"{exe(name)} missing: make -C oracle/ref_build full && make -C dropin"


def gmix(name, mode, src, dst, cwd, timeout=1100):
    r = subprocess.run([exe(name), mode, str(src), str(dst)], cwd=str(cwd), capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (name, r.stdout[-500:], r.stderr[-2000:])


def compress_pair(stock_exe, batched_exe, data, tmp_path):
    """`gmix -c` of both builds on the same bytes, side by side; returns the two working directories."""
    src = tmp_path / "input"
    src.write_bytes(data)
    dirs = []
    for name in (stock_exe, batched_exe):
        d = tmp_path / name
        d.mkdir()
        dirs.append(d)
    with ThreadPoolExecutor(2) as ex:
        list(ex.map(lambda a: gmix(a[0], "-c", src, a[1] / "c", a[1]), zip((stock_exe, batched_exe), dirs)))
    return src, dirs[0], dirs[1]


def same_outputs(stock_dir, batched_dir):
    a, b = (stock_dir / "c").read_bytes(), (batched_dir / "c").read_bytes()
    assert len(a) > 5 and a == b, "compressed bytes differ between the stock build and the run-ahead compressor"
    for t in ("entropy.tsv", "memory.tsv"):   # predictor.cpp:471-504, rows written by gmx_batched.h here
        ta, tb = (stock_dir / "analysis" / t).read_text(), (batched_dir / "analysis" / t).read_text()
        assert ta == tb, f"analysis/{t} differs"
    return a


def run_many(name, files, out_dir, chunk_bits, timeout=1100, extra=(), env=None):
    r = subprocess.run([exe(name), "-T", str(chunk_bits), *extra, str(out_dir)] + [str(f) for f in files],
                       capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])
