"""Streams of different lengths in one call (gmx_group_run_ragged, gmx_indirect_run_ragged, gmx_lstm_run_ragged): what
many files compressed side by side need when they end at different bits -- finished files leave holes among the
streams, the others go on.  The kernels with one stream per block take a count per block (one launch whatever the
lengths); every stream must come out exactly as if it had run alone."""
import numpy as np
import pytest

import goldenlib
from gmix_amd import topology

pytestmark = pytest.mark.gpu


def u32(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


@pytest.mark.parametrize("shape", ["stock", "general", "single"])
def test_mixer_group_ragged_runs_equal_the_oracle(gpu, oracle, shape):
    topo = {"stock": topology.stock(90),
            "general": topology.Topology(40, [(0, 64, 0.004)] * 5 + [(1, 16, 0.003)] * 3 + [(2, 1, 0.0005)], skip=(1,)),
            "single": topology.single(64, 256, 0.005)}[shape]
    n, m = topo.n_inputs, topo.n_mixers
    S, T = 7, 96
    g = gpu.MixerGroup(topo, S)
    b = gpu.Batch(g, T, outputs=True, mask=True)
    # ... and a batch that brings back the outputs of each stream's LAST bit only (GMX_BATCH_LAST_OUTPUTS: what a
    # compressor needs for the blackboard; the one-mixer kernel does not keep them)
    bl = None if shape == "single" else gpu.Batch(g, T, outputs=False, mask=True, last_outputs=True)
    if shape == "single":
        with pytest.raises(gpu.GmxError):
            gpu.Batch(g, T, outputs=False, mask=True, last_outputs=True)
    banks = [oracle.Bank(n, topo.skip, topo.mixers) for _ in range(S)]
    rng = np.random.default_rng(3)
    for rnd, counts in enumerate([[96, 96, 0, 17, 96, 64, 1], [0, 96, 0, 96, 5, 96, 96], [8, 8, 8, 0, 0, 0, 96]]):
        want = []
        for s in range(S):
            pred, act, ctx, bits = oracle.synth(n, m, T, seed=100 * rnd + s, ctx_mode=3, ctx_mod=5, zero_mod=4, bit_mode=1)
            b.set_records(s, pred, act, ctx, bits)
            k = counts[s]
            want.append(banks[s].run(pred[:k], act[:k], ctx[:k], bits[:k]) if k else None)
        b.upload(T)
        g.run_ragged(b, counts)
        b.download(T)
        b.wait()
        for s in range(S):
            k = counts[s]
            if k:
                assert np.array_equal(u32(b.p[s, :k]), u32(want[s][0])), (rnd, s)
                assert np.array_equal(u32(b.outputs[s, :k]), u32(want[s][1])), (rnd, s)
            assert g.export(s) == (banks[s].export_long(), banks[s].export_short()), (rnd, s)
        if bl is not None:  # one more ragged round through the other kind of batch
            counts2 = [5, 0, 96, 40, 96, 1, 96]
            want = []
            for s in range(S):
                pred, act, ctx, bits = oracle.synth(n, m, T, seed=1000 + 100 * rnd + s, ctx_mode=3, ctx_mod=5, zero_mod=4, bit_mode=1)
                bl.set_records(s, pred, act, ctx, bits)
                k = counts2[s]
                want.append(banks[s].run(pred[:k], act[:k], ctx[:k], bits[:k]) if k else None)
            bl.upload(T)
            g.run_ragged(bl, counts2)
            bl.download(T)
            bl.wait()
            for s in range(S):
                k = counts2[s]
                if k:
                    assert np.array_equal(u32(bl.p[s, :k]), u32(want[s][0])), (rnd, s)
                    assert np.array_equal(u32(bl.last_outputs[s]), u32(want[s][1][k - 1])), (rnd, s)
                assert g.export(s) == (banks[s].export_long(), banks[s].export_short()), (rnd, s)
    b.close()
    if bl is not None:
        bl.close()
    g.close()


def test_indirect_and_lstm_ragged_runs_equal_the_oracle(gpu, oracle):
    _, z = goldenlib.load("ind_tiny_dense")
    tabs = (z["ns_next"], z["rm_next"])
    models = [(256, 0.02), (3, 0.1), (65536, 0.005), (1 << 12, 0.01)]
    S, T = 5, 64
    ig = gpu.IndirectGroup(models, *tabs, S)
    ib = gpu.IndirectBatch(ig, T)
    refs = [oracle.IndirectBank(models, *tabs) for _ in range(S)]
    for rnd, counts in enumerate([[64, 0, 24, 64, 8], [0, 64, 64, 0, 16]]):
        want = []
        for s in range(S):
            ctx, bc, bits = oracle.ind_synth(len(models), T, seed=50 * rnd + s, ctx_mod=(40, 3, 0, 40))
            ib.set_records(s, ctx, bc, bits)
            k = counts[s]
            want.append(refs[s].run(ctx[:k], bc[:k], bits[:k]) if k else None)
        ib.upload(T)
        ig.run_ragged(ib, counts)
        ib.download(T)
        ib.wait()
        for s in range(S):
            k = counts[s]
            if k:
                assert np.array_equal(u32(ib.predictions[s, :k]), u32(want[s][0])), (rnd, s)
                assert np.array_equal(ib.active[s, :k], want[s][1]), (rnd, s)
            assert ig.export(s) == refs[s].export(), (rnd, s)
    ib.close()
    ig.close()
    # LSTM: whole bytes; a backward pass falls inside the longest stream only
    NB = 130
    lg = gpu.LstmGroup(4)
    lb = gpu.LstmBatch(lg, NB)
    ms = [oracle.LstmModel() for _ in range(4)]
    counts = [NB, 0, 40, 101]
    want = []
    for s in range(4):
        lg.set_weights(ms[s].weights(), stream=s)
        ppm, data = oracle.lstm_synth(NB, seed=9 + s, mask=63)
        lb.ppm[s], lb.bytes[s] = ppm, data
        k = counts[s]
        want.append(ms[s].run(ppm[:k], data[:k]) if k else None)
    lb.upload(NB)
    lg.run_ragged(lb, counts)
    lb.download(NB)
    lb.wait()
    for s in range(4):
        k = counts[s]
        if k:
            assert np.array_equal(u32(lb.predictions[s, :k]), u32(want[s][0])), s
            assert np.array_equal(lb.active[s, :k], want[s][1]) and np.array_equal(lb.contexts[s, :k], want[s][2]), s
        w, o = lg.get_weights(s)
        assert np.array_equal(u32(w), u32(ms[s].weights())) and np.array_equal(u32(o), u32(ms[s].output_layer())), s
    lb.close()
    lg.close()
