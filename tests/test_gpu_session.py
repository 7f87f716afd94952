"""The per-bit sessions (gmx_stock_session_kernel + gmx_session.inc): gmx_bank_forward /
gmx_bank_learn through a persistent wave and a mailbox must give the floats and the state the
batched kernels and the oracle give, whatever happens between two calls: the wave leaving on
its idle timer, an export, a batched run, other streams taking its slot."""
import ctypes as C
import time

import numpy as np
import pytest

from gmix_amd import topology

pytestmark = pytest.mark.gpu


def u32(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def dbg(g):
    g.L.gmx_debug_use_sessions.argtypes = [C.c_void_p, C.c_int]
    g.L.gmx_debug_open_sessions.restype = C.c_int
    g.L.gmx_debug_mailbox_on_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return g.L


def per_bit(g, rec, t0, t1, stream=0, hook=None):
    pred, act, ctx, bits = rec
    P = np.zeros(t1 - t0, np.float32)
    O = np.zeros((t1 - t0, 33), np.float32)
    for t in range(t0, t1):
        idx = np.nonzero(act[t])[0].astype(np.int32)
        P[t - t0], O[t - t0] = g.forward(pred[t], idx, ctx[t], stream=stream)
        if hook:
            hook(t, "mid")
        g.learn(int(bits[t]), stream=stream)
        if hook:
            hook(t, "end")
    return P, O


@pytest.mark.parametrize("sessions", ["device", "host", False])
def test_per_bit_equals_oracle_with_interruptions(gpu, oracle, sessions):
    """sessions: where the session's command block lives (device memory written through the BAR /
    pinned host memory), or False for two kernel launches per bit."""
    topo = topology.stock(90)
    T = 1500
    rec = oracle.synth(90, 33, T, seed=4242, ctx_mode=3, ctx_mod=5, zero_mod=7, bit_mode=1)
    ob = oracle.Bank(90, topo.skip, topo.mixers)
    p_ref, o_ref = ob.run(*rec)
    g = gpu.MixerGroup(topo, 2)
    L = dbg(g)
    assert L.gmx_debug_use_sessions(g.h, 1 if sessions else 0) == 0
    assert L.gmx_debug_mailbox_on_device(g.h, 0 if sessions == "host" else 1, 0, None) == 0
    exports = []

    def hook(t, where):
        if t % 400 == 150 and where == "mid":
            time.sleep(0.06)            # the wave leaves between Predict and Learn: replayed
        if t % 400 == 250 and where == "end":
            time.sleep(0.06)            # ... and between two bits
        if t == 700 and where == "mid":
            exports.append(g.export(0))  # stops the session, forward state must survive
        if t == 900 and where == "end":
            g.sync()
            assert g.memory_usage(3) > 0

    P, O = per_bit(g, rec, 0, T, stream=0, hook=hook)
    on_dev = C.c_int(-1)
    assert L.gmx_debug_mailbox_on_device(g.h, 0 if sessions == "host" else 1, 0, C.byref(on_dev)) == 0
    if sessions == "host":
        assert on_dev.value == 0
    assert np.array_equal(u32(O), u32(o_ref))
    assert np.array_equal(u32(P), u32(p_ref))
    assert g.export(0) == (ob.export_long(), ob.export_short())
    ob2 = oracle.Bank(90, topo.skip, topo.mixers)
    ob2.run(*[a[:700] for a in rec])
    assert exports[0] == (ob2.export_long(), ob2.export_short())
    assert L.gmx_debug_open_sessions() == 0  # the export closed it
    g.close()


def test_sessions_share_slots_and_interleave_with_batched_runs(gpu, oracle):
    topo = topology.stock(90)
    S, T = 5, 360
    recs = [oracle.synth(90, 33, T, seed=99 + s, ctx_mode=2, ctx_mod=3, zero_mod=5, bit_mode=1) for s in range(S)]
    refs = []
    for s in range(S):
        ob = oracle.Bank(90, topo.skip, topo.mixers)
        refs.append((ob,) + ob.run(*recs[s]))
    g = gpu.MixerGroup(topo, S)
    L = dbg(g)
    b = gpu.Batch(g, 120, outputs=True, mask=True)
    # bits 0..119 batched
    for s in range(S):
        b.set_records(s, *[a[:120] for a in recs[s]])
    b.upload(120)
    g.run(b, 120, learn=True)
    b.download(120)
    b.wait()
    for s in range(S):
        assert np.array_equal(u32(b.outputs[s, :120]), u32(refs[s][2][:120]))
    # bits 120..239 one at a time, all five streams in lock step: more sessions than slots,
    # every Predict of a round is done before the first Learn (so evictions hit live forwards)
    for t in range(120, 240):
        for s in range(S):
            pred, act, ctx, bits = recs[s]
            p, o = g.forward(pred[t], np.nonzero(act[t])[0].astype(np.int32), ctx[t], stream=s)
            assert np.array_equal(u32(o), u32(refs[s][2][t])), (t, s)
            assert np.float32(p).view(np.uint32) == refs[s][1][t].view(np.uint32)
        assert 1 <= L.gmx_debug_open_sessions() <= 3
        for s in range(S):
            g.learn(int(recs[s][3][t]), stream=s)
    # bits 240..359 batched again: the sessions are closed first
    for s in range(S):
        b.set_records(s, *[a[240:360] for a in recs[s]])
    b.upload(120)
    g.run(b, 120, learn=True)
    assert L.gmx_debug_open_sessions() == 0
    b.download(120)
    b.wait()
    for s in range(S):
        assert np.array_equal(u32(b.outputs[s, :120]), u32(refs[s][2][240:360]))
        assert g.export(s) == (refs[s][0].export_long(), refs[s][0].export_short())
    b.close()
    g.close()


def test_two_groups_and_protocol_errors(gpu, oracle):
    topo = topology.stock(90)
    rec = oracle.synth(90, 33, 64, seed=5, ctx_mode=3, ctx_mod=4, bit_mode=1)
    ob = oracle.Bank(90, topo.skip, topo.mixers)
    p_ref, o_ref = ob.run(*rec)
    ga, gb = gpu.MixerGroup(topo, 1), gpu.MixerGroup(topo, 1)
    with pytest.raises(gpu.GmxError):
        ga.learn(1)                      # Learn before any Predict
    for t in range(64):
        idx = np.nonzero(rec[1][t])[0].astype(np.int32)
        pa, oa = ga.forward(rec[0][t], idx, rec[2][t])
        pb, ob_ = gb.forward(rec[0][t], idx, rec[2][t])
        assert np.array_equal(u32(oa), u32(o_ref[t])) and np.array_equal(u32(ob_), u32(o_ref[t]))
        ga.learn(int(rec[3][t]))
        gb.learn(int(rec[3][t]))
        if t == 10:
            with pytest.raises(gpu.GmxError):
                ga.learn(0)              # a second Learn for the same Predict
    assert ga.export(0) == gb.export(0) == (ob.export_long(), ob.export_short())
    # copy a bank whose session is open into another group, continue there
    gc = gpu.MixerGroup(topo, 1)
    gc.copy_from(ga)
    assert gc.export(0) == ga.export(0)
    ga.close()
    gb.close()
    gc.close()


@pytest.mark.report
def test_per_bit_latency_report(gpu, capsys):
    """Not a parity test: records what a Predict+Learn pair costs through the C ABI."""
    topo = topology.stock(90)
    out = {}
    where = {}
    for sessions in (1, 2, 0):           # 1: command block on the device if the host can write there, 2: pinned host
        g = gpu.MixerGroup(topo, 1)
        L = dbg(g)
        L.gmx_debug_per_bit_latency.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        assert L.gmx_debug_use_sessions(g.h, 1 if sessions else 0) == 0
        assert L.gmx_debug_mailbox_on_device(g.h, 0 if sessions == 2 else 1, 0, None) == 0
        us = C.c_double()
        assert L.gmx_debug_per_bit_latency(g.h, 0, 500, 8, C.byref(us)) == 0
        for hold in (1, 8):
            assert L.gmx_debug_per_bit_latency(g.h, 0, 4000, hold, C.byref(us)) == 0
            out[sessions, hold] = us.value
        on_dev = C.c_int(0)
        L.gmx_debug_mailbox_on_device(g.h, 0 if sessions == 2 else 1, 0, C.byref(on_dev))
        where[sessions] = "device memory" if on_dev.value else "pinned host memory"
        g.close()
    steps = {}
    for S in (1, 64, 256, 1024):
        g = gpu.MixerGroup(topo, S)
        g.L.gmx_debug_lockstep_latency.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        us = C.c_double()
        for fused in (0, 1, 2, 3):        # bit 0: learn + next predict as one step; bit 1: persistent waves
            if (fused & 2) and S > 128:   # (the flag is ignored beyond 128 streams)
                continue
            assert g.L.gmx_debug_lockstep_latency(g.h, 100, 8, fused, C.byref(us)) == 0
            assert g.L.gmx_debug_lockstep_latency(g.h, 1000, 8, fused, C.byref(us)) == 0
            steps[S, fused] = us.value
        g.close()
    with capsys.disabled():
        print(f"\n[lock-step Predict / host round trip / Learn, C ABI] one hipGraph per half step: "
              + ", ".join(f"S = {S}: {steps[S, 0]:.1f}" for S in (1, 64, 256, 1024)) + " us/step; learn + next predict as one graph: "
              + ", ".join(f"S = {S}: {steps[S, 1]:.1f}" for S in (1, 64, 256, 1024)) + " us/step; persistent waves behind one doorbell: "
              + ", ".join(f"S = {S}: {steps[S, 2]:.1f}" for S in (1, 64)) + " us/step, learn + next predict as one command: "
              + ", ".join(f"S = {S}: {steps[S, 3]:.1f}" for S in (1, 64)) + " us/step")
        print(f"[per-bit Predict+Learn, C ABI] session, commands in {where[1]}: {out[1, 1]:.1f} us/bit (new rows "
              f"every bit), {out[1, 8]:.1f} us/bit (contexts held 8 bits); in {where[2]}: {out[2, 1]:.1f} / "
              f"{out[2, 8]:.1f}; two launches: {out[0, 1]:.1f} / {out[0, 8]:.1f} us/bit")
    # (figures are reported, not asserted: a loaded node or a busy PCIe link must not turn the parity gate red)


@pytest.mark.parametrize("shape", ["stock", "general", "stock_persistent"])
def test_lockstep_graphs_equal_oracle(gpu, oracle, shape):
    """gmx_lockstep: every stream one bit per step, each half step one hipGraph -- or, GMX_LOCKSTEP_PERSISTENT,
    persistent waves behind one doorbell that read the records from the host arrays themselves -- the floats
    and the state of the batched surface and the oracle; a batched launch in between, generation (no Learn)
    at the end, two streams that stand at different bit counts; the persistent waves also across idle exits
    (before a Predict, and between a Predict and its Learn: the forward is redone from the wave's own copy)."""
    import time
    persistent = shape.startswith("stock_persistent")
    topo = topology.stock(90) if shape != "general" else topology.Topology(
        40, [(0, 64, 0.004)] * 5 + [(1, 16, 0.003)] * 3 + [(2, 1, 0.0005)], skip=(1,))
    n, m = topo.n_inputs, topo.n_mixers
    S, T = 5, 420
    recs = [oracle.synth(n, m, T, seed=61 + s, ctx_mode=3, ctx_mod=5, zero_mod=6, bit_mode=1) for s in range(S)]
    refs = []
    for s in range(S):
        ob = oracle.Bank(n, topo.skip, topo.mixers)
        refs.append((ob,) + ob.run(*recs[s], nolearn_from=400))
    g = gpu.MixerGroup(topo, S)
    # stream 3 starts older than the others: its decay factors differ from the first step on
    pre = oracle.synth(n, m, 57, seed=5, ctx_mode=3, ctx_mod=5, bit_mode=1)
    ob3 = oracle.Bank(n, topo.skip, topo.mixers)
    ob3.run(*pre)
    g.import_(ob3.export_long(), ob3.export_short(), stream=3)
    refs[3] = (ob3,) + ob3.run(*recs[3], nolearn_from=400)
    ls = gpu.Lockstep(g, outputs=True, persistent=persistent)
    assert ls.persistent == persistent
    b = ls.batch

    def step(t, learn=True):
        for s in range(S):
            pred, act, ctx, bits = recs[s]
            b.set_records(s, pred[t:t + 1], act[t:t + 1], ctx[t:t + 1], np.zeros(1, np.uint8))
        if persistent and t == 60:
            time.sleep(0.06)                        # longer than the idle timer: the waves have left
        p = ls.predict()
        for s in range(S):
            assert np.array_equal(u32(b.outputs[s, 0]), u32(refs[s][2][t])), (t, s)
            assert np.float32(p[s]).view(np.uint32) == refs[s][1][t].view(np.uint32), (t, s)
            b.bits[s, 0] = recs[s][3][t]
        if persistent and t == 100:
            time.sleep(0.06)                        # ... between a Predict and its Learn
            b.set_records(0, recs[0][0][:1], recs[0][1][:1], recs[0][2][:1], np.zeros(1, np.uint8))  # and its record is gone
            b.bits[0, 0] = recs[0][3][t]
        if t == 120:
            # a checkpoint and a usage query between a Predict and its Learn only READ the banks: the Learn still
            # finds its Predict (the graphs' latch survives; the persistent waves leave with their forward kept)
            mid = g.export(2)
            assert g.memory_usage(0, stream=1) > 0
            ob_mid = oracle.Bank(n, topo.skip, topo.mixers)
            ob_mid.run(recs[2][0][:t], recs[2][1][:t], recs[2][2][:t], recs[2][3][:t])
            assert mid == (ob_mid.export_long(), ob_mid.export_short())
            for s in range(S):
                b.bits[s, 0] = recs[s][3][t]
        if learn:
            ls.learn()

    for t in range(0, 150):
        step(t)
    # bits 150..269 through the batched surface, then back to lock step
    bb = gpu.Batch(g, 120, outputs=True, mask=True)
    for s in range(S):
        bb.set_records(s, *[a[150:270] for a in recs[s]])
    bb.upload(120)
    g.run(bb, 120, learn=True)
    bb.download(120)
    bb.wait()
    for s in range(S):
        assert np.array_equal(u32(bb.outputs[s, :120]), u32(refs[s][2][150:270]))
    # bits 270..399: learn and the next predict as one graph
    def fill(t):
        for s in range(S):
            pred, act, ctx, bits = recs[s]
            b.set_records(s, pred[t:t + 1], act[t:t + 1], ctx[t:t + 1], np.zeros(1, np.uint8))

    fill(270)
    p = ls.predict()
    for t in range(270, 400):
        for s in range(S):
            assert np.array_equal(u32(b.outputs[s, 0]), u32(refs[s][2][t])), (t, s)
            assert np.float32(p[s]).view(np.uint32) == refs[s][1][t].view(np.uint32), (t, s)
        if t + 1 < 400:
            fill(t + 1)                              # set_records zeroes the bits ...
        for s in range(S):
            b.bits[s, 0] = recs[s][3][t]             # ... so the coded bits go in afterwards
        p = ls.learn_predict() if t + 1 < 400 else None
    ls.learn()
    with pytest.raises(gpu.GmxError):
        ls.learn()                                  # Learn twice for one Predict
    for t in range(400, 420):
        step(t, learn=False)                        # generation: Predict only
    for s in range(S):
        assert g.export(s) == (refs[s][0].export_long(), refs[s][0].export_short())
    bb.close()
    ls.close()
    g.close()
