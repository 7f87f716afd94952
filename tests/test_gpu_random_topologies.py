"""Randomised parity: topologies the fixed cases do not name (odd input counts, table sizes that
are not powers of two, several skip inputs, missing layers), random stream modes, launches cut
at random places -- the general kernel (and the single-mixer kernel where it applies) against the
oracle, bit for bit, including the checkpoint bytes."""
import numpy as np
import pytest

from gmix_amd import topology

pytestmark = pytest.mark.gpu


def u32(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def random_topology(rng):
    n = int(rng.choice([1, 2, 3, 7, 31, 64, 90, 129, 200, 256]))
    l0 = int(rng.integers(1, 7))
    l1 = int(rng.integers(0, 4))
    fin = bool(rng.integers(0, 2)) if l1 else bool(rng.integers(0, 2))
    n_skip = int(rng.integers(0, min(n, 3) + 1)) if (l1 or fin) else 0
    skip = sorted(rng.choice(n, size=n_skip, replace=False).tolist()) if n_skip else []
    sizes = [1, 2, 3, 5, 8, 100, 257, 1000, 4096]
    mixers = [(0, int(rng.choice(sizes)), float(rng.choice([0.0005, 0.003, 0.02]))) for _ in range(l0)]
    mixers += [(1, int(rng.choice(sizes)), float(rng.choice([0.0005, 0.003]))) for _ in range(l1)]
    if fin:
        mixers += [(2, int(rng.choice([1, 3, 64])), 0.001)]
    return topology.Topology(n, mixers, skip=skip)


@pytest.mark.parametrize("seed", range(16))
def test_random_topology_equals_oracle(gpu, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    topo = random_topology(rng)
    T = int(rng.integers(300, 900))
    kw = dict(ctx_mode=int(rng.integers(0, 4)), ctx_mod=int(rng.choice([1, 2, 7, 300])),
              zero_mod=int(rng.choice([0, 0, 3, 9])), bit_mode=int(rng.integers(0, 2)))
    if seed >= 12:
        kw["ctx_mode"] = 4 + (seed & 1)   # byte-held contexts with a few that move every bit
    S = 3
    streams = [oracle.synth(topo.n_inputs, topo.n_mixers, T, seed=seed * 10 + s + 1, **kw) for s in range(S)]
    g = gpu.MixerGroup(topo, S)
    use_mask = bool(rng.integers(0, 2)) or kw["zero_mod"] != 0
    cuts = sorted(set([0, T] + rng.integers(1, T, size=2).tolist()))
    chunk = max(b - a for a, b in zip(cuts, cuts[1:]))
    b = gpu.Batch(g, chunk, outputs=True, mask=use_mask)
    P = np.zeros((S, T), np.float32)
    O = np.zeros((S, T, topo.n_mixers), np.float32)
    for t0, t1 in zip(cuts, cuts[1:]):
        n = t1 - t0
        for s, (pred, act, ctx, bits) in enumerate(streams):
            b.set_records(s, pred[t0:t1], act[t0:t1], ctx[t0:t1], bits[t0:t1])
        b.upload(n)
        g.run(b, n, learn=True)
        b.download(n)
        b.wait()
        P[:, t0:t1] = b.p[:, :n]
        O[:, t0:t1] = b.outputs[:, :n]
    for s in range(S):
        ob = oracle.Bank(topo.n_inputs, topo.skip, topo.mixers)
        p_ref, o_ref = ob.run(*streams[s])
        assert np.array_equal(u32(O[s]), u32(o_ref)), (seed, s, topo.mixers, topo.skip, kw)
        assert np.array_equal(u32(P[s]), u32(p_ref))
        assert g.export(s) == (ob.export_long(), ob.export_short())
        assert [g.memory_usage(j, stream=s) for j in range(topo.n_mixers)] == [ob.memory_usage(j) for j in range(topo.n_mixers)]
    b.close()
    g.close()
