"""The register-resident throughput kernel (gmx_single.hip: banks that are one layer-0 mixer, batches
with or without an active mask) against the oracle and against the general kernel."""
import ctypes as C

import numpy as np
import pytest

from gmix_amd import topology

pytestmark = pytest.mark.gpu
GOLD = 0x9E3779B97F4A7C15


def beq(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


def set_variant(g, lanes_per_stream):
    g.L.gmx_debug_single_variant.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_single_variant(g.h, lanes_per_stream) == 0


def run_gpu(gpu, topo, streams, chunk, learn=True, force_general=False, outputs=True, variant=0, mask=False):
    S, T = len(streams), len(streams[0][3])
    g = gpu.MixerGroup(topo, S)
    set_variant(g, variant)
    if force_general:
        g.L.gmx_debug_force_general.argtypes = [C.c_void_p, C.c_int]
        g.L.gmx_debug_force_general(g.h, 1)
    b = gpu.Batch(g, chunk, outputs=outputs, mask=mask)
    P = np.zeros((S, T), np.float32)
    O = np.zeros((S, T, 1), np.float32)
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


@pytest.mark.parametrize("n,table,T,chunk,S,kw", [
    (256, 1 << 16, 3001, 3001, 1, dict()),
    (256, 1 << 16, 2000, 333, 7, dict(ctx_mode=1, ctx_mod=3)),          # rows repeat within the slot ring
    (256, 4, 2600, 1300, 5, dict(ctx_mode=3, ctx_mod=2, bit_mode=1)),   # >1024 visits of a row: shrink
    (90, 1000, 2000, 2000, 4, dict(ctx_mode=3, ctx_mod=7, zero_mod=5)),  # K=2, silent slots zeroed by caller
    (40, 17, 1500, 700, 9, dict(ctx_mode=1, ctx_mod=50, zero_mod=3, bit_mode=1)),  # K=1
    (129, 300, 1200, 1200, 3, dict(ctx_mode=1, ctx_mod=9, bit_mode=1)),  # K=4 with a ragged tail
    (64, 8, 500, 500, 2, dict(ctx_mode=1, ctx_mod=8, bit_mode=1)),       # K=1, exact fit
])
@pytest.mark.parametrize("variant", [0, 16, 32])
def test_single_kernel_matches_oracle(gpu, oracle, n, table, T, chunk, S, kw, variant):
    """variant = lanes per stream of the kernel's mapping (0: the default, 64 lanes for
    n_inputs > 128); every mapping must give the reference's floats."""
    topo = topology.single(n, table, 0.005)
    streams = [oracle.synth(n, 1, T, seed=GOLD + 31 * s, **kw) for s in range(S)]
    g, P, O = run_gpu(gpu, topo, streams, chunk, variant=variant)
    for s in range(S):
        ob = oracle.Bank(n, topo.skip, topo.mixers)
        p_ref, o_ref = ob.run(*streams[s])
        assert beq(O[s], o_ref), (s, np.argwhere(O[s].view(np.uint32) != o_ref.view(np.uint32))[:3])
        assert beq(P[s], p_ref)
        assert g.export(s) == (ob.export_long(), ob.export_short())
    g.close()


def test_single_kernel_equals_general_kernel(gpu, oracle):
    topo = topology.single(256, 1 << 12, 0.005)
    streams = [oracle.synth(256, 1, 1500, seed=5 + s, ctx_mode=1, ctx_mod=40, bit_mode=1) for s in range(6)]
    g1, P1, O1 = run_gpu(gpu, topo, streams, 500)
    g2, P2, O2 = run_gpu(gpu, topo, streams, 500, force_general=True)
    assert beq(P1, P2) and beq(O1, O2)
    for s in range(6):
        assert g1.export(s) == g2.export(s)
    g1.close()
    g2.close()


def test_single_kernel_forward_only_and_no_outputs(gpu, oracle):
    topo = topology.single(256, 64, 0.005)
    pred, act, ctx, bits = oracle.synth(256, 1, 900, ctx_mode=1, ctx_mod=64, bit_mode=1)
    ob = oracle.Bank(256, topo.skip, topo.mixers)
    p_ref, _ = ob.run(pred, act, ctx, bits, nolearn_from=600)
    g = gpu.MixerGroup(topo, 1)
    b = gpu.Batch(g, 600, outputs=False, mask=False)
    b.set_records(0, pred[:600], act[:600], ctx[:600], bits[:600])
    b.upload()
    g.run(b, 600, learn=True)
    b.download()
    b.wait()
    assert beq(b.p[0], p_ref[:600])
    before = g.export(0)
    b.set_records(0, pred[600:], act[600:], ctx[600:], bits[600:])
    b.upload(300)
    g.run(b, 300, learn=False)
    b.download(300)
    b.wait()
    assert beq(b.p[0, :300], p_ref[600:])
    assert g.export(0) == before == (ob.export_long(), ob.export_short())
    g.close()


def test_full_size_streams_sampled_against_oracle(gpu, oracle):
    """BASELINE.json configs[1] at scale: many streams with full 2^16-row tables, records
    generated on the device; a sample of streams is replayed through the oracle in full, and
    every stream must satisfy the size-independent invariants (steps_ = bits learned,
    sum of row visits = steps_, contexts_seen_ = rows with visits)."""
    topo = topology.single(256, 1 << 16, 0.005)
    S, T, rounds = 256, 512, 3
    g = gpu.MixerGroup(topo, S)
    b = gpu.Batch(g, T, outputs=False, mask=False)
    P = []
    for r in range(rounds):
        b.fill_synthetic(T, seed=777, restart=(r == 0))
        g.run(b, T)
        b.download(T)
        b.wait()
        P.append(b.p.copy())
    P = np.concatenate(P, axis=1)
    for s in (0, 1, 63, 64, 130, 255):
        seed = (777 + s * GOLD) & ((1 << 64) - 1)
        pred, act, ctx, bits = oracle.synth(256, 1, T * rounds, seed=seed)
        ob = oracle.Bank(256, topo.skip, topo.mixers)
        p_ref, _ = ob.run(pred, act, ctx, bits, want_all=False)
        assert beq(P[s], p_ref), s
        assert g.export(s) == (ob.export_long(), ob.export_short())
    for s in range(0, S, 17):
        lb, sb = g.export(s)
        steps, max_steps, seen = np.frombuffer(sb, np.uint64)
        cnt, isz = np.frombuffer(lb[:8], np.uint32)
        assert steps == T * rounds and cnt == seen and isz == 256
        rec = np.frombuffer(lb[8:], np.uint8).reshape(cnt, 12 + 1024)
        visits = rec[:, 4:12].copy().view(np.uint64).ravel()
        assert visits.sum() == steps and visits.max() == max_steps
    g.close()


@pytest.mark.parametrize("n,table,T,chunk,S,kw,variant", [
    (256, 1 << 12, 2200, 700, 5, dict(ctx_mode=3, ctx_mod=6, zero_mod=3, bit_mode=1), 0),
    (256, 4, 2600, 1300, 3, dict(ctx_mode=1, ctx_mod=2, zero_mod=2, bit_mode=1), 32),   # shrink, half the slots silent
    (90, 1000, 1500, 1500, 4, dict(ctx_mode=3, ctx_mod=7, zero_mod=5), 16),              # K = 2, ragged
    (40, 17, 900, 400, 6, dict(ctx_mode=1, ctx_mod=50, zero_mod=3, bit_mode=1), 0),      # K = 1, ragged
    (129, 300, 800, 800, 2, dict(ctx_mode=1, ctx_mod=9, zero_mod=4, bit_mode=1), 165),   # deeper ring requested
])
def test_single_kernel_with_active_mask(gpu, oracle, n, table, T, chunk, S, kw, variant):
    """Batches that carry an active_models mask: silent slots keep their stale values in the
    prediction records (the reference leaves them on the blackboard) and must neither add to the
    sum nor move their weights."""
    topo = topology.single(n, table, 0.005)
    streams = [oracle.synth(n, 1, T, seed=GOLD + 77 * s, **kw) for s in range(S)]
    assert any((st[1] == 0).any() and (st[0][st[1] == 0] != 0).any() for st in streams)   # stale, nonzero, silent
    g, P, O = run_gpu(gpu, topo, streams, chunk, variant=variant, mask=True)
    g2, P2, O2 = run_gpu(gpu, topo, streams, chunk, force_general=True, mask=True)
    assert beq(P, P2) and beq(O, O2)
    for s in range(S):
        ob = oracle.Bank(n, topo.skip, topo.mixers)
        p_ref, o_ref = ob.run(*streams[s])
        assert beq(O[s], o_ref), (s, np.argwhere(O[s].view(np.uint32) != o_ref.view(np.uint32))[:3])
        assert beq(P[s], p_ref)
        assert g.export(s) == (ob.export_long(), ob.export_short()) == g2.export(s)
    g.close()
    g2.close()
