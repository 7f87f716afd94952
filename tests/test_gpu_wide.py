"""The register-resident kernel for the 256-input 24/8/1 bank (gmx_wide.hip, BASELINE configs[2]'s
shape: two lanes per layer-0 row) against the general kernel and the oracle -- same floats, same
state, whatever the gate contexts do (new rows every bit, rows that stay, rows that come back)."""
import ctypes as C

import numpy as np
import pytest

from gmix_amd import topology

pytestmark = pytest.mark.gpu


def beq(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


def run(gpu, topo, streams, chunk, force_general, mask=True, learn_until=None):
    S, T = len(streams), len(streams[0][3])
    g = gpu.MixerGroup(topo, S)
    g.L.gmx_debug_force_general.argtypes = [C.c_void_p, C.c_int]
    g.L.gmx_debug_force_general(g.h, 1 if force_general else 0)
    b = gpu.Batch(g, chunk, outputs=True, mask=mask)
    P = np.zeros((S, T), np.float32)
    O = np.zeros((S, T, topo.n_mixers), np.float32)
    for t0 in range(0, T, chunk):
        n = min(chunk, T - t0)
        for s, (pred, act, ctx, bits) in enumerate(streams):
            b.set_records(s, pred[t0:t0 + n], act[t0:t0 + n], ctx[t0:t0 + n], bits[t0:t0 + n])
        b.upload(n)
        g.run(b, n, learn=(learn_until is None or t0 < learn_until))
        b.download(n)
        b.wait()
        P[:, t0:t0 + n] = b.p[:, :n]
        O[:, t0:t0 + n] = b.outputs[:, :n]
    b.close()
    return g, P, O


@pytest.mark.parametrize("kw,mask,table0", [
    (dict(ctx_mode=0), True, 1 << 12),                                  # every row changes every bit
    (dict(ctx_mode=3, ctx_mod=5, zero_mod=7, bit_mode=1), True, 1 << 12),  # rows persist, silent models, learnable
    (dict(ctx_mode=1, ctx_mod=2, bit_mode=1), False, 1 << 12),           # >1024 visits per row: shrink; no mask
    (dict(ctx_mode=3, ctx_mod=7, zero_mod=5, bit_mode=1), True, 1000),   # table sizes that are no powers of two
])
def test_wide_kernel_equals_general_kernel_and_oracle(gpu, oracle, kw, mask, table0):
    topo = topology.synth3(256, table0=table0, table1=(1 << 8) if table0 != 1000 else 77)
    S, T = 4, 2600
    streams = [oracle.synth(256, 33, T, seed=4321 + 13 * s, **kw) for s in range(S)]
    g1, P1, O1 = run(gpu, topo, streams, 700, force_general=False, mask=mask)
    g2, P2, O2 = run(gpu, topo, streams, 700, force_general=True, mask=mask)
    assert beq(O1, O2) and beq(P1, P2)
    for s in range(S):
        assert g1.export(s) == g2.export(s)
    ob = oracle.Bank(256, topo.skip, topo.mixers)
    p_ref, o_ref = ob.run(*streams[1])
    assert beq(O1[1], o_ref) and beq(P1[1], p_ref)
    assert g1.export(1) == (ob.export_long(), ob.export_short())
    g1.close()
    g2.close()


def test_wide_kernel_generation_tail_and_restart(gpu, oracle):
    """Predict without Learn changes nothing (runner-utils.cpp:199-209); a bank exported after a
    wide-kernel run and imported into a fresh group continues with the same floats."""
    topo = topology.synth3(256, table0=1 << 10)
    T = 1800
    st = oracle.synth(256, 33, T, seed=77, ctx_mode=3, ctx_mod=9, zero_mod=11, bit_mode=1)
    ob = oracle.Bank(256, topo.skip, topo.mixers)
    p_ref, o_ref = ob.run(*[a[:1200] for a in st])
    state = (ob.export_long(), ob.export_short())
    p_tail, o_tail = ob.run(*[a[1200:] for a in st], nolearn_from=0)
    g, P, O = run(gpu, topo, [st], 600, force_general=False, learn_until=1200)
    assert beq(P[0, :1200], p_ref) and beq(O[0, :1200], o_ref)
    assert beq(P[0, 1200:], p_tail) and beq(O[0, 1200:], o_tail)
    assert g.export(0) == state
    g2 = gpu.MixerGroup(topo, 1)
    g2.import_(*state)
    b = gpu.Batch(g2, 600, outputs=True, mask=True)
    b.set_records(0, *[a[1200:] for a in st])
    b.upload(600)
    g2.run(b, 600, learn=True)
    b.download(600)
    b.wait()
    p2, o2 = ob.run(*[a[1200:] for a in st])
    assert beq(b.p[0], p2) and beq(b.outputs[0], o2)
    assert g2.export(0) == (ob.export_long(), ob.export_short())
    b.close()
    g.close()
    g2.close()
