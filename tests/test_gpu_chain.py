"""The device-resident slice of the ensemble: LSTM byte model -> (its prediction, its context) ->
41 Indirect models -> 33 mixers, every hand-over inside HBM (gmx_lstm_feed, gmx_indirect_run's
`into`).  83 of the mixers' 90 inputs and two of the contexts never visit the host; the result must
equal the oracle chain LstmModel -> Indirect -> Mixer bit for bit."""
import numpy as np
import pytest

import goldenlib
from gmix_amd import topology

pytestmark = pytest.mark.gpu


def u32(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def test_lstm_indirect_mixer_chain_on_device(gpu, oracle):
    _, z = goldenlib.load("ind_stock41")
    tabs = (z["ns_next"], z["rm_next"])
    models = topology.stock_indirect()
    topo = topology.stock(90)
    K, N_IN, S, NB = len(models), 90, 2, 130          # 130 bytes: one LSTM backward pass inside
    T = 8 * NB
    LSTM_SLOT, IND_LSTM, MIX_LSTM = 1, 16, 22          # prediction index / context users (predictor.cpp:117, :321)
    assert topology.STOCK_INDIRECT[IND_LSTM][0] == "lstm_prediction_context"
    assert topology.STOCK_LAYER0[MIX_LSTM][0] == "lstm_prediction_context"
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    lg, ig, mg = gpu.LstmGroup(S), gpu.IndirectGroup(models, *tabs, S, slots=slots), gpu.MixerGroup(topo, S)
    lb, ib, mb = gpu.LstmBatch(lg, NB), gpu.IndirectBatch(ig, T), gpu.Batch(mg, T, outputs=True, mask=True)
    want = []
    rng = np.random.default_rng(5)
    for s in range(S):
        ppm, data = oracle.lstm_synth(NB, seed=40 + s, mask=63)
        bits = np.unpackbits(data.reshape(-1, 1), axis=1).reshape(-1)          # MSB first, like the coder
        # byte-structured contexts: bit_context = recent_bits - 1 (basic-contexts.cpp:34)
        k = np.tile(np.arange(8), NB)
        prefix = np.repeat(data.astype(np.uint32), 8) >> (8 - k)
        bc = ((1 << k) | np.where(k > 0, prefix, 0)).astype(np.uint32) - 1
        ictx = np.repeat(rng.integers(0, 5000, (NB, K)).astype(np.uint32), 8, axis=0)
        mctx = np.repeat(rng.integers(0, 1 << 16, (NB, 33)).astype(np.uint32), 8, axis=0)
        other, act_o, _, _ = oracle.synth(N_IN, 33, T, seed=70 + s, zero_mod=3)
        # ---- oracle chain
        lm = oracle.LstmModel()
        lp, la, lc = lm.run(ppm, data)
        ictx_ref, mctx_ref = ictx.copy(), mctx.copy()
        ictx_ref[:, IND_LSTM] = np.repeat(lc, 8)
        mctx_ref[:, MIX_LSTM] = np.repeat(lc, 8)
        io = oracle.IndirectBank(models, *tabs)
        ip, ia = io.run(ictx_ref, bc, bits)
        pred, act = other.copy(), np.zeros((T, N_IN), np.uint8)
        act[:, :8] = act_o[:, :8]
        pred[:, LSTM_SLOT], act[:, LSTM_SLOT] = lp.reshape(-1), la.reshape(-1)
        for i, (a, b_) in enumerate(slots):
            pred[:, a], pred[:, b_] = ip[:, 2 * i], ip[:, 2 * i + 1]
            act[:, a], act[:, b_] = ia[:, 2 * i], ia[:, 2 * i + 1]
        mo = oracle.Bank(N_IN, topo.skip, topo.mixers)
        want.append(mo.run(pred, act, mctx_ref, bits) + (mo, io, lm))
        # ---- device records: what the host still supplies
        lg.set_weights(lm.__class__().weights(), stream=s)                  # fresh reference initialisation
        lb.ppm[s], lb.bytes[s] = ppm, data
        ib.set_records(s, ictx, bc, bits)                                   # column 16 filled on the device
        act_host = np.zeros((T, N_IN), np.uint8)
        act_host[:, :8] = act_o[:, :8]
        act_host[:, LSTM_SLOT] = 0
        mb.set_records(s, other, act_host, mctx, np.zeros(T, np.uint8))     # slot 1, column 22, bits: device
    lb.upload(NB); ib.upload(T); mb.upload(T)
    lg.run(lb, NB, learn=True)
    lg.feed(lb, NB, mixer_batch=mb, slot=LSTM_SLOT, mixer_ctx_col=MIX_LSTM, ind_batch=ib, ind_ctx_col=IND_LSTM)
    ig.run(ib, T, learn=True, into=mb)
    mg.run(mb, T, learn=True)
    mb.download(T); mb.wait()
    for s in range(S):
        p_ref, o_ref, mo, io, lm = want[s]
        assert np.array_equal(u32(mb.outputs[s, :T]), u32(o_ref)), s
        assert np.array_equal(u32(mb.p[s, :T]), u32(p_ref))
        assert mg.export(s) == (mo.export_long(), mo.export_short()) and ig.export(s) == io.export()
        w, o = lg.get_weights(s)
        assert np.array_equal(u32(w), u32(lm.weights()))
    for x in (lb, ib, mb, lg, ig, mg):
        x.close()


def test_chain_over_several_steps_with_alternating_batches(gpu, oracle):
    """Three steps of the chain with two sets of downstream batches used alternately, the way
    scripts/bench_pipeline.py drives it: the uploads of step k+1 (big enough to take the transfer
    streams of their own) are queued while step k is still running; events alone keep LSTM -> feed ->
    Indirect -> mixers of one step and the bank state of consecutive steps in order."""
    _, z = goldenlib.load("ind_stock41")
    tabs = (z["ns_next"], z["rm_next"])
    models = topology.stock_indirect()
    topo = topology.stock(90)
    K, N_IN, S, NB, STEPS = len(models), 90, 8, 110, 3
    T = 8 * NB
    LSTM_SLOT, IND_LSTM, MIX_LSTM = 1, 16, 22
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    lg, ig, mg = gpu.LstmGroup(S), gpu.IndirectGroup(models, *tabs, S, slots=slots), gpu.MixerGroup(topo, S)
    # the LSTM on one half of every XCD's compute units, the other two banks on the other half: their
    # kernels then run side by side (scripts/bench_pipeline.py) -- and must give the same floats
    lg.set_cu_mask([0x0000FFFF] * 8)
    ig.set_cu_mask([0xFFFF0000] * 8)
    mg.set_cu_mask([0xFFFF0000] * 8)
    lbs = [gpu.LstmBatch(lg, NB) for _ in range(2)]
    ibs = [gpu.IndirectBatch(ig, T) for _ in range(2)]
    mbs = [gpu.Batch(mg, T, outputs=True, mask=True) for _ in range(2)]
    rng = np.random.default_rng(11)
    chains = []
    for s in range(S):
        lm = oracle.LstmModel()
        lg.set_weights(oracle.LstmModel().weights(), stream=s)
        chains.append((lm, oracle.IndirectBank(models, *tabs), oracle.Bank(N_IN, topo.skip, topo.mixers)))
    want = []

    def fill(k):
        lb, ib, mb = lbs[k & 1], ibs[k & 1], mbs[k & 1]
        refs = []
        for s in range(S):
            lm, io, mo = chains[s]
            ppm, data = oracle.lstm_synth(NB, seed=300 + 10 * k + s, mask=127)
            bits = np.unpackbits(data.reshape(-1, 1), axis=1).reshape(-1)
            kk = np.tile(np.arange(8), NB)
            prefix = np.repeat(data.astype(np.uint32), 8) >> (8 - kk)
            bc = ((1 << kk) | np.where(kk > 0, prefix, 0)).astype(np.uint32) - 1
            ictx = np.repeat(rng.integers(0, 3000, (NB, K)).astype(np.uint32), 8, axis=0)
            mctx = np.repeat(rng.integers(0, 1 << 16, (NB, 33)).astype(np.uint32), 8, axis=0)
            other, act_o, _, _ = oracle.synth(N_IN, 33, T, seed=500 + 10 * k + s, zero_mod=4)
            lp, la, lc = lm.run(ppm, data)
            ictx_ref, mctx_ref = ictx.copy(), mctx.copy()
            ictx_ref[:, IND_LSTM] = np.repeat(lc, 8)
            mctx_ref[:, MIX_LSTM] = np.repeat(lc, 8)
            ip, ia = io.run(ictx_ref, bc, bits)
            pred, act = other.copy(), np.zeros((T, N_IN), np.uint8)
            act[:, :8] = act_o[:, :8]
            pred[:, LSTM_SLOT], act[:, LSTM_SLOT] = lp.reshape(-1), la.reshape(-1)
            for i, (a, b_) in enumerate(slots):
                pred[:, a], pred[:, b_] = ip[:, 2 * i], ip[:, 2 * i + 1]
                act[:, a], act[:, b_] = ia[:, 2 * i], ia[:, 2 * i + 1]
            refs.append(mo.run(pred, act, mctx_ref, bits))
            lb.ppm[s], lb.bytes[s] = ppm, data
            ib.set_records(s, ictx, bc, bits)
            act_host = np.zeros((T, N_IN), np.uint8)
            act_host[:, :8] = act_o[:, :8]
            mb.set_records(s, other, act_host, mctx, np.zeros(T, np.uint8))
        want.append(refs)
        lb.upload(NB); ib.upload(T); mb.upload(T)

    def check(k):
        mb = mbs[k & 1]
        mb.wait()
        for s in range(S):
            p_ref, o_ref = want[k][s]
            assert np.array_equal(u32(mb.outputs[s, :T]), u32(o_ref)), (k, s)
            assert np.array_equal(u32(mb.p[s, :T]), u32(p_ref)), (k, s)

    fill(0)
    for k in range(STEPS):
        lb, ib, mb = lbs[k & 1], ibs[k & 1], mbs[k & 1]
        lg.run(lb, NB, learn=True)
        lg.feed(lb, NB, mixer_batch=mb, slot=LSTM_SLOT, mixer_ctx_col=MIX_LSTM, ind_batch=ib, ind_ctx_col=IND_LSTM)
        ig.run(ib, T, learn=True, into=mb)
        mg.run(mb, T, learn=True)
        mb.download(T)
        if k + 1 < STEPS:
            if k >= 1:
                check(k - 1)        # the other set's results are read before it is refilled
            fill(k + 1)             # host work and uploads of the next step while this one runs
    check(STEPS - 2)
    check(STEPS - 1)
    for grp in (lg, ig, mg):
        grp.set_cu_mask(None)           # back to all compute units
    for s in range(S):
        lm, io, mo = chains[s]
        assert mg.export(s) == (mo.export_long(), mo.export_short()) and ig.export(s) == io.export()
        w, o = lg.get_weights(s)
        assert np.array_equal(u32(w), u32(lm.weights()))
    for x in lbs + ibs + mbs + [lg, ig, mg]:
        x.close()
