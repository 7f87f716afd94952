"""Parity of the HIP path (through the C ABI) with the CPU oracle: every mixer output, the
clamped probability, the serialised weights and counters -- bit for bit (integer compare of
the float patterns; tolerance 0, which is stricter than BASELINE.json's 1e-6)."""
import numpy as np
import pytest

from gmix_amd import topology

pytestmark = pytest.mark.gpu

GOLD = 0x9E3779B97F4A7C15


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


def oracle_run(oracle, topo, pred, act, ctx, bits, nolearn_from=None):
    ob = oracle.Bank(topo.n_inputs, topo.skip, topo.mixers)
    p, outs = ob.run(pred, act, ctx, bits, nolearn_from=nolearn_from)
    return ob, p, outs


def gpu_run_batched(gpu, topo, streams, chunk=None, mask=True, learn=True):
    """streams: list of (pred, act, ctx, bits).  Runs all streams through one group, optionally
    in several launches of `chunk` bits (state must carry across launches)."""
    S = len(streams)
    T = len(streams[0][3])
    chunk = chunk or T
    g = gpu.MixerGroup(topo, S)
    b = gpu.Batch(g, chunk, outputs=True, mask=mask)
    P = np.zeros((S, T), np.float32)
    O = np.zeros((S, T, topo.n_mixers), np.float32)
    for t0 in range(0, T, chunk):
        n = min(chunk, T - t0)
        for s, (pred, act, ctx, bits) in enumerate(streams):
            b.set_records(s, pred[t0:t0 + n], None if act is None else act[t0:t0 + n], ctx[t0:t0 + n],
                          bits[t0:t0 + n])
        b.upload(n)
        g.run(b, n, learn=learn)
        b.download(n)
        b.wait()
        P[:, t0:t0 + n] = b.p[:, :n]
        O[:, t0:t0 + n] = b.outputs[:, :n]
    return g, P, O


CASES = [
    # name, topology, T, synth kwargs
    ("single256", lambda: topology.single(256, 1 << 16), 3000, dict()),
    ("single256_rowrepeat", lambda: topology.single(256, 1 << 16), 3000, dict(ctx_mode=1, ctx_mod=3)),
    ("single90_odd", lambda: topology.single(90, 1000), 2000, dict(ctx_mode=3, ctx_mod=7, zero_mod=5)),
    ("synth3_n90", lambda: topology.synth3(90, table0=1 << 12), 2500, dict()),
    ("synth3_n90_sticky_silent", lambda: topology.synth3(90, table0=1 << 12), 2500,
     dict(ctx_mode=3, ctx_mod=5, zero_mod=7)),
    ("synth3_n256", lambda: topology.synth3(256, table0=1 << 10), 1500, dict(ctx_mode=1, ctx_mod=50)),
    ("stock90_learnable", lambda: topology.stock(90), 4000, dict(ctx_mode=2, zero_mod=9, bit_mode=1)),
    ("stock90_smallctx", lambda: topology.stock(90), 3000, dict(ctx_mode=1, ctx_mod=2, bit_mode=1)),
    ("n7_l0_3_l1_2", lambda: topology.Topology(7, [(0, 5, .02), (0, 1, .01), (0, 300, .03), (1, 2, .01),
                                                   (1, 9, .02), (2, 1, .005)], skip=(0, 3)), 3000,
     dict(ctx_mode=1, ctx_mod=11, zero_mod=3, bit_mode=1)),
    ("no_final_no_skip", lambda: topology.Topology(33, [(0, 64, .01)] * 5 + [(1, 8, .01)] * 2, skip=()),
     2000, dict(ctx_mode=1, ctx_mod=70, bit_mode=1)),
]


@pytest.mark.parametrize("name,mk,T,kw", CASES, ids=[c[0] for c in CASES])
def test_batched_matches_oracle(gpu, oracle, name, mk, T, kw):
    topo = mk()
    pred, act, ctx, bits = oracle.synth(topo.n_inputs, topo.n_mixers, T, **kw)
    ob, p_ref, o_ref = oracle_run(oracle, topo, pred, act, ctx, bits)
    g, P, O = gpu_run_batched(gpu, topo, [(pred, act, ctx, bits)])
    assert bits_equal(O[0], o_ref), f"{name}: mixer outputs differ at bit {np.argwhere(O[0].view(np.uint32) != o_ref.view(np.uint32))[:3]}"
    assert bits_equal(P[0], p_ref)
    lg, sg = g.export(0)
    assert sg == ob.export_short()
    assert lg == ob.export_long()
    for j in (0, topo.n_mixers - 1):
        assert g.memory_usage(j) == ob.memory_usage(j)
    g.close()


def test_chunked_launches_carry_state(gpu, oracle):
    topo = topology.stock(90)
    T = 3000
    pred, act, ctx, bits = oracle.synth(90, 33, T, ctx_mode=3, ctx_mod=300, zero_mod=6, bit_mode=1)
    ob, p_ref, o_ref = oracle_run(oracle, topo, pred, act, ctx, bits)
    g, P, O = gpu_run_batched(gpu, topo, [(pred, act, ctx, bits)], chunk=257)
    assert bits_equal(O[0], o_ref) and bits_equal(P[0], p_ref)
    assert g.export(0) == (ob.export_long(), ob.export_short())
    g.close()


def test_many_streams_independent(gpu, oracle):
    topo = topology.synth3(90, table0=1 << 8)
    S, T = 37, 600
    streams = [oracle.synth(90, 33, T, seed=GOLD + 977 * s, ctx_mode=1, ctx_mod=40 + s, zero_mod=4)
               for s in range(S)]
    g, P, O = gpu_run_batched(gpu, topo, streams)
    for s in (0, 1, 17, 36):
        ob, p_ref, o_ref = oracle_run(oracle, topo, *streams[s])
        assert bits_equal(O[s], o_ref) and bits_equal(P[s], p_ref), f"stream {s}"
        assert g.export(s) == (ob.export_long(), ob.export_short())
    g.close()


def test_maskless_batch_equals_masked(gpu, oracle):
    """Without a mask the caller zeroes silent slots; exact because they add nothing."""
    topo = topology.single(256, 1 << 10)
    T = 1500
    pred, act, ctx, bits = oracle.synth(256, 1, T, ctx_mode=1, ctx_mod=20, zero_mod=3)
    ob, p_ref, o_ref = oracle_run(oracle, topo, pred, act, ctx, bits)
    g, P, O = gpu_run_batched(gpu, topo, [(pred, act, ctx, bits)], mask=False)
    assert bits_equal(O[0], o_ref) and bits_equal(P[0], p_ref)
    g.close()


def test_forward_only_leaves_state(gpu, oracle):
    """Generation mode: Predict+Perceive without Learn (runner-utils.cpp:199-209) -- the long
    state must not change (tester.cpp:358-366)."""
    topo = topology.stock(90)
    T = 1200
    pred, act, ctx, bits = oracle.synth(90, 33, T, ctx_mode=3, ctx_mod=9, bit_mode=1)
    ob, p_ref, o_ref = oracle_run(oracle, topo, pred, act, ctx, bits, nolearn_from=800)
    g = gpu.MixerGroup(topo, 1)
    b = gpu.Batch(g, T)
    b.set_records(0, pred[:800], act[:800], ctx[:800], bits[:800])
    b.upload(800)
    g.run(b, 800, learn=True)
    b.download(800)
    b.wait()
    assert bits_equal(b.p[0, :800], p_ref[:800])
    before = g.export(0)
    b.set_records(0, pred[800:], act[800:], ctx[800:], bits[800:])
    b.upload(400)
    g.run(b, 400, learn=False)
    b.download(400)
    b.wait()
    assert bits_equal(b.p[0, :400], p_ref[800:]) and bits_equal(b.outputs[0, :400], o_ref[800:])
    assert g.export(0) == before == (ob.export_long(), ob.export_short())
    g.close()


def test_per_bit_surface_matches_batched(gpu, oracle):
    """Predict()/Perceive()/Learn() one bit at a time gives the same floats as the batch."""
    topo = topology.stock(90)
    T = 300
    pred, act, ctx, bits = oracle.synth(90, 33, T, ctx_mode=3, ctx_mod=4, zero_mod=8, bit_mode=1)
    ob, p_ref, o_ref = oracle_run(oracle, topo, pred, act, ctx, bits)
    g = gpu.MixerGroup(topo, 2)
    for t in range(T):
        idx = np.nonzero(act[t])[0].astype(np.int32)
        p, out = g.forward(pred[t], idx, ctx[t], stream=1)
        assert bits_equal(out, o_ref[t]) and np.float32(p).view(np.uint32) == p_ref[t].view(np.uint32), t
        g.learn(bits[t], stream=1)
    assert g.export(1) == (ob.export_long(), ob.export_short())
    fresh = oracle.Bank(90, topo.skip, topo.mixers)
    assert g.export(0) == (fresh.export_long(), fresh.export_short())  # stream 0 untouched
    with pytest.raises(gpu.GmxError):
        g.learn(1, stream=1)  # Learn without a preceding Predict
    g.close()


def test_export_import_copy_roundtrip(gpu, oracle):
    """The reference's restart tests (tester.cpp:330-348): checkpoint half way, restore into a
    fresh bank (or Copy), finish -- identical output and identical re-serialisation."""
    topo = topology.stock(90)
    T = 2000
    pred, act, ctx, bits = oracle.synth(90, 33, T, ctx_mode=2, zero_mod=5, bit_mode=1)
    ob, p_ref, o_ref = oracle_run(oracle, topo, pred, act, ctx, bits)
    h = T // 2
    g1, P1, O1 = gpu_run_batched(gpu, topo, [(pred[:h], act[:h], ctx[:h], bits[:h])])
    lg, sg = g1.export(0)
    g2 = gpu.MixerGroup(topo, 3)
    g2.import_(lg, sg, stream=2)
    g2.copy_from(g1, 0, 1)
    assert g2.export(2) == (lg, sg) and g2.export(1) == (lg, sg)
    b = gpu.Batch(g2, T - h)
    for s in range(3):
        b.set_records(s, pred[h:], act[h:], ctx[h:], bits[h:])
    b.upload()
    g2.run(b)
    b.download()
    b.wait()
    for s in (1, 2):
        assert bits_equal(b.p[s], p_ref[h:]) and bits_equal(b.outputs[s], o_ref[h:])
        assert g2.export(s) == (ob.export_long(), ob.export_short())
    assert not bits_equal(b.p[0], p_ref[h:])  # the fresh bank really is different
    g1.close()
    g2.close()


@pytest.mark.parametrize("kw", [dict(ctx_mode=3, ctx_mod=17, zero_mod=6, bit_mode=1),
                                dict(ctx_mode=5, ctx_mod=300, zero_mod=6, bit_mode=1),
                                dict(ctx_mode=4, bit_mode=1)],
                         ids=["byte_held_mod", "bitlevel_mod", "bitlevel"])
def test_device_synth_fill_matches_host_generator(gpu, oracle, kw):
    """The on-device record generator is oracle/gmx_synth.h restated: same stream per seed."""
    topo = topology.stock(90) if kw["ctx_mode"] == 4 else topology.synth3(90, table0=1 << 8)
    S, T = 5, 400
    g = gpu.MixerGroup(topo, S)
    b = gpu.Batch(g, T, outputs=True, mask=True)
    b.fill_synthetic(200, seed=12345, restart=True, **kw)
    g.run(b, 200)
    b.download(200)
    b.wait()
    P = b.p[:, :200].copy()
    b.fill_synthetic(200, seed=12345, restart=False, **kw)   # continue the same streams
    g.run(b, 200)
    b.download(200)
    b.wait()
    P = np.concatenate([P, b.p[:, :200]], axis=1)
    for s in range(S):
        seed = (12345 + s * GOLD) & ((1 << 64) - 1)
        pred, act, ctx, bits = oracle.synth(90, 33, T, seed=seed, **kw)
        ob, p_ref, _ = oracle_run(oracle, topo, pred, act, ctx, bits)
        assert bits_equal(P[s], p_ref), f"stream {s}"
        assert g.export(s) == (ob.export_long(), ob.export_short())
    g.close()


def test_compressed_bytes_identical(gpu, oracle):
    """What the north star asks for: the arithmetic coder fed with GPU probabilities writes the
    same bytes as with the reference's probabilities."""
    topo = topology.stock(90)
    T = 8000
    pred, act, ctx, bits = oracle.synth(90, 33, T, ctx_mode=3, ctx_mod=64, zero_mod=10, bit_mode=1)
    _, p_ref, _ = oracle_run(oracle, topo, pred, act, ctx, bits)
    g, P, _ = gpu_run_batched(gpu, topo, [(pred, act, ctx, bits)], chunk=1000)
    a, b = oracle.encode(bits, P[0]), oracle.encode(bits, p_ref)
    assert a == b and len(a) < T // 8  # identical, and the learnable stream really compresses
    g.close()


def test_double_buffered_batches_keep_their_order(gpu, oracle):
    """Two record batches used alternately the way BASELINE configs[3] asks for: while the kernel
    works on one, the other's records cross PCIe on the upload stream and the previous results come
    back on the download stream.  Events, not host waits, keep upload -> run -> download of a batch
    and run(k) -> run(k+1) of the bank in order; the floats must be the oracle's."""
    from gmix_amd import topology
    topo = topology.stock(90)
    S, T, chunk = 6, 2400, 300
    streams = [oracle.synth(90, 33, T, seed=31 + 7 * s, ctx_mode=3, ctx_mod=6, zero_mod=9, bit_mode=1) for s in range(S)]
    g = gpu.MixerGroup(topo, S)
    bs = [gpu.Batch(g, chunk, outputs=True, mask=True) for _ in range(2)]
    P = np.zeros((S, T), np.float32)
    O = np.zeros((S, T, 33), np.float32)

    def fill(b, k):
        for s, (pred, act, ctx, bits) in enumerate(streams):
            sl = slice(k * chunk, (k + 1) * chunk)
            b.set_records(s, pred[sl], act[sl], ctx[sl], bits[sl])

    n = T // chunk
    fill(bs[0], 0)
    bs[0].upload(chunk)
    for k in range(n):
        cur, nxt = bs[k & 1], bs[(k + 1) & 1]
        g.run(cur, chunk, learn=True)
        if k + 1 < n:
            nxt.wait()                        # results of chunk k-1 are on the host ...
            if k >= 1:
                P[:, (k - 1) * chunk:k * chunk] = nxt.p[:, :chunk]
                O[:, (k - 1) * chunk:k * chunk] = nxt.outputs[:, :chunk]
            fill(nxt, k + 1)                  # ... so its host arrays may be refilled
            nxt.upload(chunk)                 # beside the kernel of chunk k
        cur.download(chunk)
    for k in (n - 2, n - 1):
        b = bs[k & 1]
        b.wait()
        P[:, k * chunk:(k + 1) * chunk] = b.p[:, :chunk]
        O[:, k * chunk:(k + 1) * chunk] = b.outputs[:, :chunk]
    for s in range(S):
        ob = oracle.Bank(90, topo.skip, topo.mixers)
        p_ref, o_ref = ob.run(*streams[s])
        assert np.array_equal(O[s].view(np.uint32), o_ref.view(np.uint32)), s
        assert np.array_equal(P[s].view(np.uint32), p_ref.view(np.uint32)), s
        assert g.export(s) == (ob.export_long(), ob.export_short())
    for b in bs:
        b.close()
    g.close()
