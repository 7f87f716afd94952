"""Shared by tests/test_batched_cpu.py and tests/test_gpu_batched.py: the run-ahead compressor
(gmix_amd/host/gmx_batched.h inside builds of the reference, oracle/ref_build/Makefile `batched`) beside
the stock build of the reference on the same bytes."""
import json
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


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


def need(*exes):
    for exe in exes:
        assert os.path.exists(os.path.join(REF, exe)), f"oracle/_ref/{exe} missing: make -C oracle/ref_build batched"


def gmix(exe, mode, src, dst, cwd, timeout=1100):
    r = subprocess.run([os.path.join(REF, exe), mode, str(src), str(dst)], cwd=str(cwd), capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (exe, r.stdout[-500:], r.stderr[-2000:])


def compress_pair(stock_exe, batched_exe, data, tmp_path):
    """`gmix -c` of both builds on the same bytes, side by side; returns the two working directories."""
    src = tmp_path / "input"
    src.write_bytes(data)
    dirs = []
    for exe in (stock_exe, batched_exe):
        d = tmp_path / exe
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


def run_many(exe, files, out_dir, chunk_bits, timeout=1100, extra=()):
    r = subprocess.run([os.path.join(REF, exe), "-T", str(chunk_bits), *extra, str(out_dir)] + [str(f) for f in files],
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def train_pair(stock_exe, batched_exe, train, test, tmp_path, checkpoints=None, env=None, timeout=1100):
    """`gmix -t [checkpoint] train test` (runner_utils::RunTraining, runner-utils.cpp:222-322) of both builds side by
    side, each in a directory of its own; returns the two directories."""
    ftrain, ftest = tmp_path / "train", tmp_path / "test"
    ftrain.write_bytes(train)
    ftest.write_bytes(test)
    dirs = []
    for k, exe in enumerate((stock_exe, batched_exe)):
        d = tmp_path / f"t{len(list(tmp_path.iterdir()))}_{exe}"
        d.mkdir()
        dirs.append(d)

    def one(k):
        exe, d = (stock_exe, batched_exe)[k], dirs[k]
        args = [os.path.join(REF, exe), "-t"] + ([str(checkpoints[k])] if checkpoints else []) + [str(ftrain), str(ftest)]
        r = subprocess.run(args, cwd=str(d), capture_output=True, text=True, timeout=timeout,
                           env=dict(os.environ, **(env or {})))
        assert r.returncode == 0 and "training cross entropy" in r.stdout, (exe, r.stdout[-500:], r.stderr[-2000:])
        return r.stdout[r.stdout.index("training cross entropy"):].splitlines()[0]

    with ThreadPoolExecutor(2) as ex:
        said = list(ex.map(one, range(2)))
    assert said[0] == said[1], said
    return dirs


def same_training(stock_dir, batched_dir):
    """What RunTraining leaves: data/tmp (the coded training file), analysis/training.tsv (train and test cross
    entropy every other per cent), the two analysis tables, and data/trained_checkpoint."""
    from dropin_common import same_checkpoint
    for f in ("data/tmp", "analysis/training.tsv", "analysis/entropy.tsv", "analysis/memory.tsv"):
        a, b = (stock_dir / f).read_bytes(), (batched_dir / f).read_bytes()
        assert len(a) > 0 and a == b, f"{f} differs"
    same_checkpoint(str(stock_dir / "data" / "trained_checkpoint"), str(batched_dir / "data" / "trained_checkpoint"))
