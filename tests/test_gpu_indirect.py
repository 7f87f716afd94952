"""The Indirect-model banks (gmx_indirect.hip behind gmx_indirect_* of include/gmxmix.h) against
the golden vectors of the REAL reference class and against the oracle: every prediction slot and
active flag, the checkpoint bytes, per-bit vs batched, and the models feeding a mixer batch
inside HBM."""
import numpy as np
import pytest

import goldenlib
from gmix_amd import topology
from golden.cases import IND_CASES

pytestmark = pytest.mark.gpu


def u32(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def run_batched(gpu, models, tabs, streams, chunk, nolearn_from=None):
    """streams: [(ctx, bc, bits)]; returns (group, pred[S,T,2K], act[S,T,2K])."""
    S, T, K = len(streams), len(streams[0][2]), len(models)
    g = gpu.IndirectGroup(models, tabs[0], tabs[1], S)
    b = gpu.IndirectBatch(g, chunk)
    P = np.zeros((S, T, 2 * K), np.float32)
    A = np.zeros((S, T, 2 * K), np.uint8)
    t0 = 0
    while t0 < T:
        n = min(chunk, T - t0)
        learn = True
        if nolearn_from is not None:
            if t0 < nolearn_from:
                n = min(n, nolearn_from - t0)
            else:
                learn = False
        for s, (ctx, bc, bits) in enumerate(streams):
            b.set_records(s, ctx[t0:t0 + n], bc[t0:t0 + n], bits[t0:t0 + n])
        b.upload(n)
        g.run(b, n, learn=learn)
        b.download(n)
        b.wait()
        P[:, t0:t0 + n] = b.predictions[:, :n]
        A[:, t0:t0 + n] = b.active[:, :n]
        t0 += n
    b.close()
    return g, P, A


@pytest.mark.parametrize("name", sorted(IND_CASES))
def test_indirect_kernel_matches_reference_goldens(gpu, oracle, name):
    meta, models, ctx, bc, bits, nolearn, z = goldenlib.ind_case(name)
    T = meta["T"]
    chunk = 4096 if T > 50000 else 1000
    g, P, A = run_batched(gpu, models, (z["ns_next"], z["rm_next"]), [(ctx, bc, bits)], chunk, nolearn)
    D, K = meta["dump"], len(models)
    if D:
        assert np.array_equal(u32(P[0, :D]), z["pred"])
        assert np.array_equal(A[0, :D], np.unpackbits(z["active"], axis=1, bitorder="little")[:, :2 * K])
    assert oracle.ind_fnv64(P[0], A[0]) == meta["h64"]
    e = g.export(0)
    assert len(e) == meta["long_len"] and goldenlib.sha256(e) == meta["long_sha256"]
    assert [g.memory_usage(i) for i in range(K)] == meta["usage"]
    g.close()


def test_indirect_streams_are_independent_and_per_bit_agrees(gpu, oracle):
    _, z = goldenlib.load("ind_tiny_dense")
    tabs = (z["ns_next"], z["rm_next"])
    models = [(256, 0.02), (3, 0.1), (4096, 0.005), (1, 0.5), (65536, 0.02)]
    S, T = 4, 1500
    streams = [oracle.ind_synth(len(models), T, seed=11 + s, ctx_mod=(40, 3, 900, 0)) for s in range(S)]
    g, P, A = run_batched(gpu, models, tabs, streams, 700)
    refs = []
    for s in range(S):
        ob = oracle.IndirectBank(models, *tabs)
        p, a = ob.run(*streams[s])
        assert np.array_equal(u32(P[s]), u32(p)) and np.array_equal(A[s], a), s
        assert g.export(s) == ob.export()
        refs.append(ob)
    # continue stream 1 one bit at a time (Predict / Learn), stream 2 untouched
    ctx, bc, bits = oracle.ind_synth(len(models), 300, seed=99, ctx_mod=(40, 3, 900, 0))
    for t in range(300):
        p, a = g.forward(ctx[t], bc[t], stream=1)
        pr, ar = refs[1].predict(ctx[t], bc[t])
        assert np.array_equal(u32(p), u32(pr)) and np.array_equal(a, ar), t
        if t % 7 != 3:                       # a Predict without Learn now and then (generation)
            g.learn(bits[t], stream=1)
            refs[1].learn(bits[t])
    assert g.export(1) == refs[1].export() and g.export(2) == refs[2].export()
    with pytest.raises(gpu.GmxError):
        g.learn(1, stream=2)                 # Learn without Predict
    # checkpoint round trip and copy
    g2 = gpu.IndirectGroup(models, *tabs, 2)
    g2.import_(g.export(1), stream=0)
    g2.copy_from(g, src_stream=3, dst_stream=1)
    assert g2.export(0) == g.export(1) and g2.export(1) == g.export(3)
    with pytest.raises(gpu.GmxError):
        g2.import_(g.export(1)[:-5], stream=0)
    g.close()
    g2.close()


@pytest.mark.parametrize("sessions", [1, 0])
def test_indirect_per_bit_surface_sessions_and_launches(gpu, oracle, sessions):
    """gmx_indirect_forward / gmx_indirect_learn through the persistent session (one command per bit, the
    learn riding with the next forward) and through a kernel launch per call: the oracle's floats either way
    -- with Predicts that are never learned, with the wave leaving on its idle timer between a forward and
    its learn (it comes back and recomputes the forward from the intact payload slot) and between bits,
    with a batched run in between (the session writes its logit tables back first), export at the end."""
    import ctypes as C
    import time
    _, z = goldenlib.load("ind_tiny_dense")
    tabs = (z["ns_next"], z["rm_next"])
    models = [(256, 0.02), (3, 0.1), (4096, 0.005), (1, 0.5), (65536, 0.02), (1 << 15, 1.0 / 200)]
    g = gpu.IndirectGroup(models, *tabs, 2)
    g.L.gmx_debug_indirect_use_sessions.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_indirect_use_sessions(g.h, sessions) == 0
    refs = [oracle.IndirectBank(models, *tabs) for _ in range(2)]
    ctx, bc, bits = oracle.ind_synth(len(models), 420, seed=5, ctx_mod=(40, 3, 900, 0))
    for t in range(420):
        s = 1 if 300 <= t < 340 else 0       # a second stream gets a session of its own for a while
        p, a = g.forward(ctx[t], bc[t], stream=s)
        pr, ar = refs[s].predict(ctx[t], bc[t])
        assert np.array_equal(u32(p), u32(pr)) and np.array_equal(a, ar), t
        if t in (10, 310):                   # the path under test is the one in use
            assert (g.L.gmx_debug_open_sessions() > 0) == bool(sessions)
        if t == 50:
            time.sleep(0.05)                 # > the session's idle timer: the wave leaves holding a forward
        if t % 9 != 4:
            g.learn(bits[t], stream=s)
            refs[s].learn(bits[t])
        if t == 120:
            time.sleep(0.05)                 # ... and between bits, with the learn only noted so far
        if t == 200:                         # batched traffic on the same banks in between
            b = gpu.IndirectBatch(g, 64)
            c2, b2, x2 = oracle.ind_synth(len(models), 64, seed=77, ctx_mod=(40, 3, 900, 0))
            for k in range(2):
                b.set_records(k, c2, b2, x2)
                refs[k].run(c2, b2, x2)
            b.upload()
            g.run(b)
            b.download()
            b.wait()
            b.close()
    for k in range(2):
        assert g.export(k) == refs[k].export(), k
    g.close()


def test_indirect_models_feed_the_mixers_inside_hbm(gpu, oracle):
    """41 stock Indirect models -> 82 of the 90 mixer inputs, written by gmx_indirect_run straight
    into the mixer batch's device records; the other 8 inputs, the mixer contexts come from the
    host.  The mixer outputs must equal the oracle chain Indirect -> Mixer."""
    _, z = goldenlib.load("ind_stock41")
    tabs = (z["ns_next"], z["rm_next"])
    models = topology.stock_indirect()
    K, N, T, S = len(models), 90, 1200, 2
    topo = topology.stock(90)
    # blackboard layout as in the reference's constructor order: 8 other predictions first
    # (slots 0..7: PPMd, LSTM, matches), then the Indirect pairs
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    ig = gpu.IndirectGroup(models, *tabs, S, slots=slots)
    mg = gpu.MixerGroup(topo, S)
    ib = gpu.IndirectBatch(ig, T)
    mb = gpu.Batch(mg, T, outputs=True, mask=True)
    want = []
    for s in range(S):
        ctx, bc, bits = oracle.ind_synth(K, T, seed=500 + s, ctx_mod=(300, 0, 70000, 5))
        other, act_o, mctx, _ = oracle.synth(N, 33, T, seed=900 + s, ctx_mode=2, zero_mod=4)
        ib.set_records(s, ctx, bc, bits)
        act_full = np.zeros((T, N), np.uint8)
        act_full[:, :8] = act_o[:, :8]
        mb.set_records(s, other, act_full, mctx, np.zeros(T, np.uint8))  # bits arrive from the models' batch
        # oracle chain
        ob = oracle.IndirectBank(models, *tabs)
        ip, ia = ob.run(ctx, bc, bits)
        pred = other.copy()
        act = act_full.copy()
        for i, (a, b_) in enumerate(slots):
            pred[:, a], pred[:, b_] = ip[:, 2 * i], ip[:, 2 * i + 1]
            act[:, a], act[:, b_] = ia[:, 2 * i], ia[:, 2 * i + 1]
        om = oracle.Bank(N, topo.skip, topo.mixers)
        want.append(om.run(pred, act, mctx, bits) + (om, ob))
    ib.upload(T)
    mb.upload(T)
    ig.run(ib, T, learn=True, into=mb)
    mg.run(mb, T, learn=True)
    mb.download(T)
    mb.wait()
    for s in range(S):
        p_ref, o_ref, om, ob = want[s]
        assert np.array_equal(u32(mb.outputs[s, :T]), u32(o_ref)), s
        assert np.array_equal(u32(mb.p[s, :T]), u32(p_ref))
        assert mg.export(s) == (om.export_long(), om.export_short())
        assert ig.export(s) == ob.export()
    for x in (ib, mb, ig, mg):
        x.close()


@pytest.mark.parametrize("mode", ["sessions", "no_indirect_session", "no_sessions"])
def test_chain_forward_one_round_trip_per_bit(gpu, oracle, mode):
    """gmx_chain_forward: the Indirect models' Predict and the mixers' Predict of a bit as one call -- with both
    banks on per-bit sessions the Indirect wave hands its 82 predictions to the mixers' wave and rings its
    mailbox itself -- against the oracle chain Indirect -> Mixer, bit by bit with Learn, a Predict without Learn
    now and then, idle exits of both sessions, and with the fallbacks (a launch per call on either side)."""
    import ctypes as C
    import time
    _, z = goldenlib.load("ind_stock41")
    tabs = (z["ns_next"], z["rm_next"])
    models = topology.stock_indirect()
    K, N, T = len(models), 90, 260
    topo = topology.stock(90)
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    ig = gpu.IndirectGroup(models, *tabs, 1, slots=slots)
    mg = gpu.MixerGroup(topo, 1)
    ig.L.gmx_debug_indirect_use_sessions.argtypes = [C.c_void_p, C.c_int]
    mg.L.gmx_debug_use_sessions.argtypes = [C.c_void_p, C.c_int]
    assert ig.L.gmx_debug_indirect_use_sessions(ig.h, 0 if mode != "sessions" else 1) == 0
    assert mg.L.gmx_debug_use_sessions(mg.h, 0 if mode == "no_sessions" else 1) == 0
    ctx, bc, bits = oracle.ind_synth(K, T, seed=321, ctx_mod=(300, 0, 70000, 5))
    other, act_o, mctx, _ = oracle.synth(N, 33, T, seed=654, ctx_mode=4, zero_mod=4)
    ob = oracle.IndirectBank(models, *tabs)
    om = oracle.Bank(N, topo.skip, topo.mixers)
    for t in range(T):
        if t in (90, 170):
            time.sleep(0.06)                     # both sessions' waves have left (170: after a Predict nobody learned)
        ip, ia = ob.predict(ctx[t], bc[t])
        pred = other[t].copy()
        act = np.zeros(N, np.uint8)
        act[:8] = act_o[t, :8]
        host_active = np.flatnonzero(act).astype(np.int32)   # the blackboard without the Indirect models
        for i, (a, b_) in enumerate(slots):
            pred[a], pred[b_] = ip[2 * i], ip[2 * i + 1]
            act[a], act[b_] = ia[2 * i], ia[2 * i + 1]
        p_ref, o_ref = om.predict(pred, np.flatnonzero(act), mctx[t])
        stale = other[t].copy()                   # whatever stands in the Indirect slots must not matter
        stale[8:] = 123.0
        p, out, gp, ga = ig.chain_forward(mg, ctx[t], bc[t], stale, host_active, mctx[t])
        assert np.array_equal(u32(gp), u32(ip)) and np.array_equal(ga, ia), t
        assert np.array_equal(u32(out), u32(o_ref)) and np.float32(p).view(np.uint32) == np.float32(p_ref).view(np.uint32), t
        if t % 40 != 39:
            ig.learn(bits[t])
            mg.learn(bits[t])
            ob.learn(bits[t])
            om.learn(bits[t])
    assert ig.export(0) == ob.export()
    assert mg.export(0) == (om.export_long(), om.export_short())
    ig.close()
    mg.close()


def test_chain_forward_into_a_wide_bank(gpu, oracle):
    """gmx_chain_forward with mixers that are not the stock shape (256 inputs, the Indirect models' slots beyond
    input 128): the two calls with the host in between, same floats as the oracle chain."""
    _, z = goldenlib.load("ind_tiny_dense")
    tabs = (z["ns_next"], z["rm_next"])
    models = [(256, 0.02), (65536, 0.02), (32768, 0.005)]
    K, N, T = len(models), 256, 48
    topo = topology.synth3(N, table0=1 << 8)
    slots = [(250, 255), (128, 131), (5, 200)]
    ig = gpu.IndirectGroup(models, *tabs, 1, slots=slots)
    mg = gpu.MixerGroup(topo, 1)
    ctx, bc, bits = oracle.ind_synth(K, T, seed=77, ctx_mod=(50, 0, 3000, 7))
    other, act_o, mctx, _ = oracle.synth(N, topo.n_mixers, T, seed=78, ctx_mode=1, ctx_mod=97, zero_mod=5)
    ob = oracle.IndirectBank(models, *tabs)
    om = oracle.Bank(N, topo.skip, topo.mixers)
    own = sorted(x for ab in slots for x in ab)
    for t in range(T):
        ip, ia = ob.predict(ctx[t], bc[t])
        pred = other[t].copy()
        act = act_o[t].astype(np.uint8).copy()
        host_active = np.flatnonzero(act).astype(np.int32)   # lists the Indirect slots too: they are ignored
        act[own] = 0
        for i, (a, b_) in enumerate(slots):
            pred[a], pred[b_] = ip[2 * i], ip[2 * i + 1]
            act[a], act[b_] = ia[2 * i], ia[2 * i + 1]
        p_ref, o_ref = om.predict(pred, np.flatnonzero(act), mctx[t])
        p, out, gp, ga = ig.chain_forward(mg, ctx[t], bc[t], other[t], host_active, mctx[t])
        assert np.array_equal(u32(gp), u32(ip)) and np.array_equal(ga, ia), t
        assert np.array_equal(u32(out), u32(o_ref)) and np.float32(p).view(np.uint32) == np.float32(p_ref).view(np.uint32), t
        ig.learn(bits[t]); mg.learn(bits[t]); ob.learn(bits[t]); om.learn(bits[t])
    assert ig.export(0) == ob.export()
    assert mg.export(0) == (om.export_long(), om.export_short())
    ig.close()
    mg.close()


def test_indirect_device_synth_is_the_oracles_stream(gpu, oracle):
    _, z = goldenlib.load("ind_tiny_dense")
    models = [(256, 0.02), (65536, 0.02), (32768, 0.005)]
    S, T = 3, 900
    g = gpu.IndirectGroup(models, z["ns_next"], z["rm_next"], S)
    b = gpu.IndirectBatch(g, T)
    b.fill_synthetic(500, seed=1234, restart=True, ctx_mod=(50, 0, 3000, 7))
    g.run(b, 500, learn=True)
    b.download(500)
    b.wait()
    first = (b.predictions[:, :500].copy(), b.active[:, :500].copy())
    b.fill_synthetic(400, restart=False, ctx_mod=(50, 0, 3000, 7))   # the stream continues
    g.run(b, 400, learn=True)
    b.download(400)
    b.wait()
    for s in range(S):
        ctx, bc, bits = oracle.ind_synth(len(models), T, seed=(1234 + s * 0x9E3779B97F4A7C15) % (1 << 64),
                                         ctx_mod=(50, 0, 3000, 7))
        ob = oracle.IndirectBank(models, z["ns_next"], z["rm_next"])
        p, a = ob.run(ctx, bc, bits)
        assert np.array_equal(u32(first[0][s]), u32(p[:500])) and np.array_equal(first[1][s], a[:500])
        assert np.array_equal(u32(b.predictions[s, :400]), u32(p[500:])) and np.array_equal(b.active[s, :400], a[500:])
        assert g.export(s) == ob.export()
        # the blackboard slots as the bank holds them: the last bit's (a silent model repeats its slot)
        assert np.array_equal(u32(g.slot_values(s)), u32(p[-1]))
    # ... and as a caller sets them: silent models report what was set
    v = np.arange(1, 7, dtype=np.float32)
    g.set_slot_values(v, 1)
    assert np.array_equal(g.slot_values(1), v) and not np.array_equal(g.slot_values(0), v)
    g.reset()                                                # every state "never seen": all six slots stay silent
    g.set_slot_values(v, 1)
    pr, ac = g.forward(np.array([1, 2, 3], np.uint32), 1, stream=1)
    assert np.array_equal(pr, v) and not ac.any()
    b.close()
    g.close()


@pytest.mark.parametrize("lrs", [(0.02, 1.0, 0.005), (0.02, 1.5, 0.005), (3.0, 0.02, 0.7)])
def test_indirect_logistic_short_and_general_way(gpu, oracle, lrs):
    """The batched kernel takes the logistic's short instruction sequence for a whole launch when every logit of
    the stream starts below 32 in magnitude and no learning rate exceeds 1 (then none can ever reach 64, where the
    short way ends), and the general function otherwise: learning rates at 1, and beyond (logits then move in steps
    larger than 1 and the general function runs), byte-shaped records, long enough for logits to saturate."""
    _, z = goldenlib.load("ind_tiny_dense")
    models = [(256, lrs[0]), (65536, lrs[1]), (32768, lrs[2])]
    S, T = 2, 6000
    g = gpu.IndirectGroup(models, z["ns_next"], z["rm_next"], S)
    b = gpu.IndirectBatch(g, T)
    b.fill_synthetic(T, seed=4321, restart=True, ctx_mod=(20, 0, 300, 7))   # few contexts: states are revisited
    g.run(b, T, learn=True)
    b.download(T)
    b.wait()
    for s in range(S):
        ctx, bc, bits = oracle.ind_synth(len(models), T, seed=(4321 + s * 0x9E3779B97F4A7C15) % (1 << 64),
                                         ctx_mod=(20, 0, 300, 7))
        ob = oracle.IndirectBank(models, z["ns_next"], z["rm_next"])
        p, a = ob.run(ctx, bc, bits)
        assert np.array_equal(u32(b.predictions[s, :T]), u32(p)) and np.array_equal(b.active[s, :T], a)
        assert g.export(s) == ob.export()
    b.close()
    g.close()


@pytest.mark.parametrize("shape", ["random", "shifted_bytes", "near_contexts", "repeated_bytes"])
def test_indirect_block_pipeline_on_irregular_records(gpu, oracle, shape):
    """The batched kernel fetches a block's table entries one block ahead and patches what the block in
    between wrote.  Records that are not the reference's byte-shaped ones take its other paths: entries that
    repeat inside a block and across blocks, bit_context >= 256, blocks that straddle bytes, contexts whose
    index ranges overlap, a launch length that is no multiple of the block."""
    _, z = goldenlib.load("ind_tiny_dense")
    tabs = (z["ns_next"], z["rm_next"])
    models = [(256, 0.02), (3, 0.1), (4096, 0.005), (1, 0.5), (65536, 0.02), (7, 0.05)]
    K, T, S = len(models), 1237, 3
    streams = []
    for s in range(S):
        rng = np.random.default_rng(77 + s)
        bits = rng.integers(0, 2, T).astype(np.uint8)
        if shape == "random":  # nothing byte-like: few contexts, any bit_context
            ctx = rng.integers(0, 5, (T, K)).astype(np.uint32)
            bc = rng.integers(0, 300, T).astype(np.uint32)
        else:
            nb = T // 8 + 2
            byte_ctx = rng.integers(0, 3 if shape == "repeated_bytes" else 50, (nb, K)).astype(np.uint32)
            if shape == "near_contexts":  # context c and c+1 of a 1-entry ... tables: (c << 8) % size wraps into its neighbours
                byte_ctx = (rng.integers(0, 2, (nb, K)) + 0xffffff).astype(np.uint32)
            byte_val = rng.integers(0, 2 if shape == "repeated_bytes" else 256, nb)
            off = 3 if shape == "shifted_bytes" else 0  # blocks of 8 start 3 bits into a byte
            ctx = np.zeros((T, K), np.uint32)
            bc = np.zeros(T, np.uint32)
            for t in range(T):
                by, j = divmod(t + off, 8)
                ctx[t] = byte_ctx[by]
                bc[t] = (1 << j) | (int(byte_val[by]) >> (8 - j))
                bits[t] = (int(byte_val[by]) >> (7 - j)) & 1
        streams.append((ctx, bc, bits))
    g, P, A = run_batched(gpu, models, tabs, streams, 1237)
    for s in range(S):
        ob = oracle.IndirectBank(models, *tabs)
        p, a = ob.run(*streams[s])
        assert np.array_equal(u32(P[s]), u32(p)) and np.array_equal(A[s], a), (shape, s)
        assert g.export(s) == ob.export()
    g.close()
