"""BASELINE.json configs[1] at the bench's full stream count (4096 streams, 258 GiB of gate
tables; 512 bits per launch here, bench.py times 1024) and the 256-input 24/8/1 bank at bench scale
(1024 streams: one wave on every SIMD; bench.py runs two) through properties that do not need the oracle to run 1.5 M bits: a launch split in two
gives the same floats and the same banks; sampled streams equal the oracle run on the same seed;
streams do not leak into each other."""
import numpy as np
import pytest

from gmix_amd import topology

pytestmark = pytest.mark.gpu

S, T = 4096, 512   # bench.py's stream count for --config single; half its bits per launch
GOLD = 0x9E3779B97F4A7C15
SEED = 0x1234567


def u32(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def test_full_size_run_is_split_invariant_and_sampled_streams_match_the_oracle(gpu, oracle):
    topo = topology.single(256, 1 << 16, 0.005)
    try:
        g = gpu.MixerGroup(topo, S)
    except gpu.GmxError as e:  # a box with less free HBM than 258 GiB cannot host this case
        pytest.skip(f"cannot allocate {S} banks: {e}")
    b = gpu.Batch(g, T, outputs=False, mask=False)
    samples = [0, 1, 777, 2048, S - 1]
    _split_invariance_and_samples(gpu, oracle, topo, g, b, S, samples)


def test_wide_bank_at_bench_size(gpu, oracle):
    """gmx_wide_kernel at bench.py's synth3 shape: 256 inputs x 24/8/1, 2^12-row layer-0 tables,
    1024 streams x 512 bits, new gate rows every bit."""
    topo = topology.synth3(256, table0=1 << 12)
    Sw = 1024
    g = gpu.MixerGroup(topo, Sw)
    b = gpu.Batch(g, T, outputs=False, mask=False)
    _split_invariance_and_samples(gpu, oracle, topo, g, b, Sw, [0, 1, 500, Sw - 1])


def _split_invariance_and_samples(gpu, oracle, topo, g, b, S, samples):

    def run(splits):
        g.reset()
        b.fill_synthetic(T, seed=SEED, restart=True)
        # the device generator continues a stream across calls, so generate once, run in pieces
        # by pointing successive launches at the same records: a split is two batches
        if splits == 1:
            g.run(b, T, learn=True)
            b.download(T)
            b.wait()
            return b.p.copy()
        half = gpu.Batch(g, T // 2, outputs=False, mask=False)
        out = np.zeros((S, T), np.float32)
        for k in range(2):
            half.fill_synthetic(T // 2, seed=SEED, restart=(k == 0))
            g.run(half, T // 2, learn=True)
            half.download(T // 2)
            half.wait()
            out[:, k * (T // 2):(k + 1) * (T // 2)] = half.p
        half.close()
        return out

    p1 = run(1)
    banks1 = {s: g.export(s) for s in samples}
    p2 = run(2)
    assert np.array_equal(u32(p1), u32(p2))
    for s in samples:
        assert g.export(s) == banks1[s]
    # every probability is a clamped logistic
    assert np.isfinite(p1).all() and p1.min() >= np.float32(1e-4) and p1.max() <= np.float32(1) - np.float32(1e-4)
    # sampled streams against the oracle, from the seed alone
    for s in samples:
        pred, act, ctx, bits = oracle.synth(256, topo.n_mixers, T, seed=(SEED + s * GOLD) % (1 << 64))
        ob = oracle.Bank(256, topo.skip, topo.mixers)
        p_ref, _ = ob.run(pred, act, ctx, bits)
        assert np.array_equal(u32(p1[s]), u32(p_ref)), s
        assert banks1[s] == (ob.export_long(), ob.export_short()), s
    # distinct seeds give distinct streams (no stream reads another's rows or records)
    assert len({p1[s].tobytes() for s in samples}) == len(samples)
    b.close()
    g.close()
