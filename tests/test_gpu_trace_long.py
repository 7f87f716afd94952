"""Real feature-model inputs at length on the MI355X (VERDICT r1 "real-data parity is thin"): the
reference's own Predictor (oracle/_ref/ref_trace: the reference compiled in the build container; the
binary travels) records the mixer boundary of every bit of a text and of a binary that exist on
both boxes; the HIP batched path -- gmx_stock_kernel, the kernel BASELINE configs[2..4] run -- replays
all of them in chunks like a compressor would and must give every output of every mixer, every
probability, the final .long / .short bytes and the arithmetic-coded bytes."""
import os

import numpy as np
import pytest

from gmix_amd.bank import Topology
from trace_common import REF_TRACE, ROOT, make_trace

pytestmark = pytest.mark.gpu

# GMX_CORPUS=/path/to/enwik8 replays the named data's first bytes as well (BASELINE.json configs[0] / [2]); the fixed
# cases below use files that are on both boxes whatever else is
CASES = {
    # name: (file present on both boxes, bytes, analysis, chunk bits)
    "text_30k": (os.path.join(ROOT, "SURVEY.md"), 30000, 0, 16384),
    "text_12k_analysis": (os.path.join(ROOT, "SURVEY.md"), 12000, 1, 5000),
    "binary_20k": (REF_TRACE, 20000, 0, 65536),
}
if os.environ.get("GMX_CORPUS"):
    CASES["corpus_100k"] = (os.environ["GMX_CORPUS"], 100000, 0, 65536)


@pytest.mark.parametrize("name", sorted(CASES))
def test_batched_kernels_replay_the_reference_predictor(gpu, oracle, tmp_path, name):
    src, n_bytes, analysis, chunk = CASES[name]
    assert os.path.exists(REF_TRACE), "oracle/_ref/ref_trace missing: make -C oracle/ref_build full"
    tr = make_trace(src, n_bytes, str(tmp_path / "t.bin"), analysis)
    T = tr["T"]
    assert T == 8 * n_bytes and tr["n"] == 90 and tr["M"] == 33
    topo = Topology(tr["n"], tr["mixers"], tr["skip"])
    g = gpu.MixerGroup(topo, 1)
    b = gpu.Batch(g, chunk, outputs=True, mask=True)
    p = np.zeros(T, np.float32)
    for t0 in range(0, T, chunk):
        k = min(chunk, T - t0)
        b.set_records(0, tr["pred"][t0:t0 + k], tr["act"][t0:t0 + k], tr["ctx"][t0:t0 + k], tr["bits"][t0:t0 + k])
        b.upload(k)
        g.run(b, k)
        b.download(k)
        b.wait()
        bad = np.nonzero((b.outputs[0, :k].view(np.uint32) != tr["outs"][t0:t0 + k].view(np.uint32)).any(axis=1))[0]
        assert len(bad) == 0, f"first differing bit {t0 + bad[0]}"
        p[t0:t0 + k] = b.p[0, :k]
    assert np.array_equal(p.view(np.uint32), tr["p"].view(np.uint32))
    lb, sb = g.export(0)
    assert lb == tr["long"] and sb == tr["short"]
    assert oracle.encode(tr["bits"], p) == oracle.encode(tr["bits"], tr["p"])
    b.close()
    g.close()
