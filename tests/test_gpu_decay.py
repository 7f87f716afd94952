"""The bit-count factor of the learning-rate decay, float(0.9 / pow(1e-7 * steps + 0.8, 0.8))
(mixer.cpp:111), when the device makes the table (gmx_decay_kernel: launches that cover streams at
many different bit counts): the same floats as the libm the reference calls, entry for entry."""
import ctypes as C

import numpy as np
import pytest

from gmix_amd import topology

pytestmark = pytest.mark.gpu


def host_table(oracle, steps0, T):
    L = oracle.lib()
    L.gmxo_decay_base.restype = C.c_float
    L.gmxo_decay_base.argtypes = [C.c_uint64]
    out = np.zeros((len(steps0), T), np.float32)
    for u, s0 in enumerate(steps0):
        for t in range(T):
            out[u, t] = L.gmxo_decay_base(int(s0) + t)
    return out


def test_device_decay_table_equals_libm(gpu, oracle):
    g = gpu.MixerGroup(topology.single(64, 16, 0.005), 1)
    g.L.gmx_debug_decay_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(5)
    # starts across the whole range a stream can reach, the first bits, the crossing of 2^32, a duplicate
    steps0 = np.concatenate([[0, 1, (1 << 32) - 300, 12345, 12345], rng.integers(0, 1 << 40, 43, dtype=np.uint64)]).astype(np.uint64)
    T = 4096
    out = np.zeros((len(steps0), T), np.float32)
    n = C.c_uint32(0)
    assert g.L.gmx_debug_decay_table(g.h, steps0.ctypes.data_as(C.c_void_p), len(steps0), T,
                                     out.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
    ref = host_table(oracle, steps0, T)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert n.value < 64          # a double within 2^-46 of a float rounding boundary is rare
    g.close()


def test_streams_at_different_bit_counts_in_one_launch(gpu, oracle):
    """Banks restored from checkpoints of different ages, then one batched launch over all of them:
    the decay table has one row per age and comes from the device."""
    topo = topology.stock(90)
    S, T = 24, 400                                 # 24 rows x 400 entries: above the device threshold
    g = gpu.MixerGroup(topo, S)
    refs = []
    for s in range(S):
        ob = oracle.Bank(90, topo.skip, topo.mixers)
        pre = oracle.synth(90, 33, 31 * s + 5, seed=900 + s, ctx_mode=3, ctx_mod=4, bit_mode=1)
        ob.run(*pre)
        g.import_(ob.export_long(), ob.export_short(), stream=s)
        refs.append(ob)
    b = gpu.Batch(g, T, outputs=True, mask=True)
    streams = [oracle.synth(90, 33, T, seed=77 + s, ctx_mode=3, ctx_mod=4, zero_mod=6, bit_mode=1) for s in range(S)]
    for s in range(S):
        b.set_records(s, *streams[s])
    b.upload(T)
    g.run(b, T, learn=True)
    b.download(T)
    b.wait()
    for s in range(S):
        p_ref, o_ref = refs[s].run(*streams[s])
        assert np.array_equal(b.outputs[s].view(np.uint32), o_ref.view(np.uint32)), s
        assert np.array_equal(b.p[s].view(np.uint32), p_ref.view(np.uint32)), s
        assert g.export(s) == (refs[s].export_long(), refs[s].export_short())
    b.close()
    g.close()
