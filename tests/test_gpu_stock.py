"""The register-resident kernel for the reference's own shape (gmx_stock.hip: 90 inputs,
24/8/1, one skip input) against the general kernel and the oracle -- same floats, same state."""
import ctypes as C

import numpy as np
import pytest

from gmix_amd import topology

pytestmark = pytest.mark.gpu


def beq(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


def run(gpu, topo, streams, chunk, force_general, mask=True, learn=True, exact=False, pairs=False, staged=-1, outputs=True):
    S, T = len(streams), len(streams[0][3])
    g = gpu.MixerGroup(topo, S)
    g.L.gmx_debug_stock_staged.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_stock_staged(g.h, staged) == 0
    g.L.gmx_debug_stock_pairs.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_stock_pairs(g.h, 1 if pairs else 0) == 0
    g.L.gmx_debug_force_general.argtypes = [C.c_void_p, C.c_int]
    g.L.gmx_debug_force_general(g.h, 1 if force_general else 0)
    g.L.gmx_debug_stock_exact.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_stock_exact(g.h, 1 if exact else 0) == 0
    b = gpu.Batch(g, chunk, outputs=outputs, mask=mask)
    P = np.zeros((S, T), np.float32)
    O = np.zeros((S, T, topo.n_mixers), np.float32)
    for t0 in range(0, T, chunk):
        n = min(chunk, T - t0)
        for s, (pred, act, ctx, bits) in enumerate(streams):
            b.set_records(s, pred[t0:t0 + n], act[t0:t0 + n], ctx[t0:t0 + n], bits[t0:t0 + n])
        b.upload(n)
        g.run(b, n, learn=learn)
        b.download(n)
        b.wait()
        P[:, t0:t0 + n] = b.p[:, :n]
        if outputs:
            O[:, t0:t0 + n] = b.outputs[:, :n]
    return g, P, O


@pytest.mark.parametrize("kw,mask", [
    (dict(ctx_mode=0), True),                                  # every row changes every bit
    (dict(ctx_mode=3, ctx_mod=5, zero_mod=7, bit_mode=1), True),  # rows persist, silent models, learnable
    (dict(ctx_mode=1, ctx_mod=2, bit_mode=1), False),           # >1024 visits per row: shrink; no mask
    (dict(ctx_mode=4, zero_mod=11, bit_mode=1), True),          # a real run's pattern: 4 rows move every bit (the sparse path)
    (dict(ctx_mode=5, ctx_mod=3, bit_mode=1), False),           # ... between a handful of rows: write-back, refetch, shrink
])
def test_stock_kernel_equals_general_kernel_and_oracle(gpu, oracle, kw, mask):
    topo = topology.stock(90)
    S, T = 5, 2600
    streams = [oracle.synth(90, 33, T, seed=1234 + 17 * s, **kw) for s in range(S)]
    g1, P1, O1 = run(gpu, topo, streams, 700, force_general=False, mask=mask, staged=1)   # rows through the LDS images
    g2, P2, O2 = run(gpu, topo, streams, 700, force_general=True, mask=mask)
    g3, P3, O3 = run(gpu, topo, streams, 700, force_general=False, mask=mask, staged=0)   # rows lane-private
    assert beq(P1, P2) and beq(O1, O2) and beq(P1, P3) and beq(O1, O3)
    for s in range(S):
        assert g1.export(s) == g2.export(s) == g3.export(s)
    ob = oracle.Bank(90, topo.skip, topo.mixers)
    p_ref, o_ref = ob.run(*streams[2])
    assert beq(O1[2], o_ref) and beq(P1[2], p_ref)
    assert g1.export(2) == (ob.export_long(), ob.export_short())
    # the masked forward chains (the fallback for non-finite values) give the same floats
    g3, P3, O3 = run(gpu, topo, streams, 700, force_general=False, mask=mask, exact=True)
    assert beq(P1, P3) and beq(O1, O3)
    for s in range(S):
        assert g1.export(s) == g3.export(s)
    # Predict + Learn with only the probabilities stored: the kernel's "plain" build (mode tests folded at
    # compile time), rows through the LDS images and lane-private
    for staged in (1, 0):
        g5, P5, _ = run(gpu, topo, streams, 700, force_general=False, mask=mask, staged=staged, outputs=False)
        assert beq(P1, P5), staged
        for s in range(S):
            assert g1.export(s) == g5.export(s), (staged, s)
        g5.close()
    # the lane-pair kernel (gmx_wide.hip instantiated for 90 inputs) gives the same floats
    g4, P4, O4 = run(gpu, topo, streams, 700, force_general=False, mask=mask, pairs=True)
    assert beq(P1, P4) and beq(O1, O4)
    for s in range(S):
        assert g1.export(s) == g4.export(s)
    g1.close()
    g2.close()
    g3.close()
    g4.close()


def test_stock_kernel_non_finite_values_stay_where_the_reference_puts_them(gpu, oracle):
    """An infinite input reaches every mixer whose row exists -- and only those: a mixer on a row
    it has never learned still predicts exactly 0 (mixer.cpp:52-55).  The fast chains would
    smear NaN over such lanes through their zero weights; the kernel must notice and redo the
    bit with the masked chains."""
    topo = topology.stock(90)
    T = 260
    pred, act, ctx, bits = oracle.synth(90, 33, T, seed=909, ctx_mode=3, ctx_mod=4, bit_mode=1)
    pred = pred.copy()
    act = act.copy()
    ctx = ctx.copy()
    pred[200, 7] = np.inf           # an active input blows up at bit 200 ...
    act[200, 7] = 1
    ctx[200:, 3] = 0x12345          # ... while mixer 3 moves to a row it has never seen
    ob = oracle.Bank(90, topo.skip, topo.mixers)
    p_ref, o_ref = ob.run(pred[:201], act[:201], ctx[:201], bits[:201], nolearn_from=200)
    g, P, O = run(gpu, topo, [(pred[:200], act[:200], ctx[:200], bits[:200])], 200, False)
    assert beq(O[0], o_ref[:200])
    idx = np.nonzero(act[200])[0].astype(np.int32)
    p, out = g.forward(pred[200], idx, ctx[200])
    assert out[3] == 0.0 and o_ref[200][3] == 0.0
    fin = np.isfinite(o_ref[200])
    assert np.array_equal(np.isfinite(out), fin) and not fin.all()
    assert beq(out[fin], o_ref[200][fin])
    g.close()


def test_stock_kernel_plain_build_redoes_a_bit_with_non_finite_values(gpu, oracle):
    """The same through the batched Predict + Learn launch that stores probabilities only (the "plain" build):
    an infinite active input in the middle of a run, with and without active masks; the probabilities equal the
    general kernel's, the banks afterwards those of the build with the mode tests."""
    topo = topology.stock(90)
    T = 300
    for mask in (True, False):
        pred, act, ctx, bits = oracle.synth(90, 33, T, seed=911, ctx_mode=3, ctx_mod=4, bit_mode=1)
        pred, act, ctx = pred.copy(), act.copy(), ctx.copy()
        pred[150, 7] = np.inf
        act[150, 7] = 1
        ctx[150:, 3] = 0x12345      # mixer 3 on a row it has never seen while the others blow up
        st = [(pred, act, ctx, bits)]
        ga, Pa, _ = run(gpu, topo, st, 100, force_general=True, mask=mask)
        for staged in (1, 0):
            gs, Ps, _ = run(gpu, topo, st, 100, force_general=False, mask=mask, staged=staged)
            gb, Pb, _ = run(gpu, topo, st, 100, force_general=False, mask=mask, staged=staged, outputs=False)
            assert np.array_equal(Pa.view(np.uint32), Pb.view(np.uint32)), (mask, staged)
            assert np.array_equal(Ps.view(np.uint32), Pb.view(np.uint32)), (mask, staged)
            # the same instruction streams with and without the mode tests: the same bytes, NaN payloads included
            # (against the general kernel the NaNs a learn step leaves in the weights differ in payload only)
            assert gs.export(0) == gb.export(0), (mask, staged)
            gs.close()
            gb.close()
        ga.close()


def test_stock_kernel_forward_only_and_per_bit(gpu, oracle):
    topo = topology.stock(90)
    T = 500
    pred, act, ctx, bits = oracle.synth(90, 33, T, seed=77, ctx_mode=3, ctx_mod=6, zero_mod=9, bit_mode=1)
    ob = oracle.Bank(90, topo.skip, topo.mixers)
    p_ref, o_ref = ob.run(pred, act, ctx, bits, nolearn_from=400)
    g, P, O = run(gpu, topo, [(pred[:400], act[:400], ctx[:400], bits[:400])], 400, False)
    assert beq(P[0], p_ref[:400])
    before = g.export(0)
    for t in range(400, T):  # Predict without Learn through the per-bit surface
        idx = np.nonzero(act[t])[0].astype(np.int32)
        p, out = g.forward(pred[t], idx, ctx[t])
        assert beq(out, o_ref[t]) and np.float32(p).view(np.uint32) == p_ref[t].view(np.uint32), t
    assert g.export(0) == before
    g.close()


def test_staged_rows_with_every_simd_busy(gpu, oracle):
    """The staged row path at the stream counts it is chosen for (>= 512: every SIMD of the chip holds a
    wave, four per CU share an LDS): 1024 streams x 96 bits with new rows every bit and byte-held rows mixed
    across streams, against the lane-private path on the same records (same floats, same banks) and, for
    sampled streams, the oracle."""
    topo = topology.stock(90)
    S, T = 1024, 96
    outs = {}
    for staged in (1, 0):
        g = gpu.MixerGroup(topo, S)
        g.L.gmx_debug_stock_staged.argtypes = [C.c_void_p, C.c_int]
        assert g.L.gmx_debug_stock_staged(g.h, staged) == 0
        b = gpu.Batch(g, T, outputs=False, mask=False)
        for rep in range(3):  # later passes revisit rows the first one wrote back; the third moves 4 rows per bit
            b.fill_synthetic(T, seed=4242, restart=True, ctx_mode=(2, 0, 4)[rep], ctx_mod=1)
            g.run(b, T)
        b.download(T)
        b.wait()
        outs[staged] = (b.p.copy(), [g.export(s) for s in (0, 1, 511, 777, S - 1)])
        b.close()
        g.close()
    assert np.array_equal(outs[1][0].view(np.uint32), outs[0][0].view(np.uint32))
    assert outs[1][1] == outs[0][1]
    GOLD = 0x9E3779B97F4A7C15
    for k, s in enumerate((0, 1, 511, 777, S - 1)):
        ob = oracle.Bank(90, topo.skip, topo.mixers)
        seed = (4242 + s * GOLD) % (1 << 64)
        ob.run(*oracle.synth(90, 33, T, seed=seed, ctx_mode=2, ctx_mod=1))
        ob.run(*oracle.synth(90, 33, T, seed=seed, ctx_mode=0, ctx_mod=1))
        p_ref, _ = ob.run(*oracle.synth(90, 33, T, seed=seed, ctx_mode=4, ctx_mod=1))
        assert np.array_equal(outs[1][0][s].view(np.uint32), p_ref.view(np.uint32)), s
        assert outs[1][1][k] == (ob.export_long(), ob.export_short()), s
