"""The HIP path against the golden vectors recorded from the REFERENCE itself (not merely
against the oracle): synthetic cases replayed from their seeds, and the english.dic trace of
the whole reference Predictor replayed from its recorded mixer-boundary inputs."""
import numpy as np
import pytest

import goldenlib
from golden.cases import CASES

pytestmark = pytest.mark.gpu


def run_gpu_case(gpu, oracle, meta, chunk=4096, want_outputs=True):
    topo = goldenlib.topo_of(meta)
    kw, nolearn = goldenlib.synth_kwargs(meta)
    T = meta["T"]
    st = oracle.Stream(topo.n_inputs, topo.n_mixers, **kw)
    g = gpu.MixerGroup(topo, 1)
    b = gpu.Batch(g, chunk, outputs=True, mask=True)
    h = 0
    outs_d, p_d = [], []
    for t0 in range(0, T, chunk):
        n = min(chunk, T - t0)
        pred, act, ctx, bits = st.next(n)
        b.set_records(0, pred, act, ctx, bits)
        b.upload(n)
        if nolearn is not None and t0 < nolearn < t0 + n:
            raise AssertionError("chunk must not straddle nolearn_from")
        g.run(b, n, learn=(nolearn is None or t0 < nolearn))
        b.download(n)
        b.wait()
        h = oracle.fnv64(b.outputs[0, :n], b.p[0, :n], h0=h)
        if t0 < meta["dump"]:
            outs_d.append(b.outputs[0, :n].copy())
            p_d.append(b.p[0, :n].copy())
    return g, h, outs_d, p_d


SHORT = [n for n, c in CASES.items() if c[1] <= 10000]
LONG = [n for n, c in CASES.items() if c[1] > 10000]


def check(gpu, oracle, name, chunk):
    meta, z = goldenlib.load(name)
    g, h, outs_d, p_d = run_gpu_case(gpu, oracle, meta, chunk)
    assert h == meta["h64"], f"{name}: checksum over all {meta['T']} bits differs from the reference"
    d = meta["dump"]
    if d:
        assert np.array_equal(np.concatenate(outs_d)[:d].view(np.uint32), z["outs"])
        assert np.array_equal(np.concatenate(p_d)[:d].view(np.uint32), z["p"])
    lb, sb = g.export(0)
    assert sb.hex() == meta["short_hex"]
    assert len(lb) == meta["long_len"] and goldenlib.sha256(lb) == meta["long_sha256"]
    assert [g.memory_usage(j) for j in range(len(meta["mixers"]))] == list(z["mem"])
    g.close()


@pytest.mark.parametrize("name", SHORT)
def test_gpu_matches_reference_short(gpu, oracle, name):
    chunk = 800 if name == "stock90_nolearn_tail" else 1000
    check(gpu, oracle, name, chunk)


@pytest.mark.parametrize("name", LONG)
def test_gpu_matches_reference_long(gpu, oracle, name):
    check(gpu, oracle, name, 20000)


def test_gpu_matches_reference_predictor_trace(gpu, oracle):
    """Real feature-model inputs (whole reference Predictor on english.dic): outputs,
    probabilities, serialised state and the arithmetic-coded bytes."""
    meta, z = goldenlib.load("trace_english")
    pred, act, ctx, bits, outs_ref, p_ref = goldenlib.unpack_trace(z, meta)
    topo = goldenlib.topo_of(meta)
    T = meta["T"]
    g = gpu.MixerGroup(topo, 1)
    b = gpu.Batch(g, T, outputs=True, mask=True)
    b.set_records(0, pred, act, ctx, bits)
    b.upload()
    g.run(b)
    b.download()
    b.wait()
    assert np.array_equal(b.outputs[0].view(np.uint32), outs_ref)
    assert np.array_equal(b.p[0].view(np.uint32), p_ref)
    lb, sb = g.export(0)
    assert sb.hex() == meta["short_hex"]
    assert len(lb) == meta["long_len"] and goldenlib.sha256(lb) == meta["long_sha256"]
    assert oracle.encode(bits, b.p[0]) == oracle.encode(bits, p_ref.view(np.float32))
    # and bit by bit through Predict()/Learn()
    g2 = gpu.MixerGroup(topo, 1)
    for t in range(T):
        idx = np.nonzero(act[t])[0].astype(np.int32)
        p, out = g2.forward(pred[t], idx, ctx[t])
        assert np.array_equal(out.view(np.uint32), outs_ref[t]), t
        assert np.float32(p).view(np.uint32) == p_ref[t]
        g2.learn(bits[t])
    assert g2.export(0) == (lb, sb)
    # ... and through the lock-step graphs (the surface for many decoders at once)
    g3 = gpu.MixerGroup(topo, 2)
    ls = gpu.Lockstep(g3, outputs=True)
    for t in range(T):
        for s in range(2):
            ls.batch.set_records(s, pred[t:t + 1], act[t:t + 1], ctx[t:t + 1], bits[t:t + 1])
        p = ls.predict()
        assert np.array_equal(ls.batch.outputs[1, 0].view(np.uint32), outs_ref[t]), t
        assert p[0].view(np.uint32) == p_ref[t] and p[1].view(np.uint32) == p_ref[t]
        ls.learn()
    assert g3.export(0) == (lb, sb) and g3.export(1) == (lb, sb)
    ls.close()
    g.close()
    g2.close()
    g3.close()
