"""gmx_chainstep: S decoders in lock step through the LSTM byte model, the 41 Indirect models and the 33 mixers --
one device step (one hipGraph) per coded bit, the bit of a step known only when its probability is back
(coder/decoder.cpp:19-39).  Must equal the oracle chain LstmModel -> Indirect -> Mixer bit for bit, like the batched
chain of tests/test_gpu_chain.py, whose record recipe this file shares."""
import numpy as np
import pytest

import goldenlib
from gmix_amd import topology

pytestmark = pytest.mark.gpu

LEARN, PREDICT = 1, 2


def u32(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def chain_stream(oracle, models, tabs, topo, slots, NB, seed, rng, lstm_slot=1, ind_lstm=16, mix_lstm=22):
    """One stream's records and what the oracle chain makes of them."""
    K, N_IN, T = len(models), 90, 8 * NB
    ppm, data = oracle.lstm_synth(NB, seed=seed, mask=63)
    bits = np.unpackbits(data.reshape(-1, 1), axis=1).reshape(-1)
    k = np.tile(np.arange(8), NB)
    prefix = np.repeat(data.astype(np.uint32), 8) >> (8 - k)
    bc = ((1 << k) | np.where(k > 0, prefix, 0)).astype(np.uint32) - 1    # bit_context = recent_bits - 1
    ictx = np.repeat(rng.integers(0, 5000, (NB, K)).astype(np.uint32), 8, axis=0)
    mctx = np.repeat(rng.integers(0, 1 << 16, (NB, 33)).astype(np.uint32), 8, axis=0)
    other, act_o, _, _ = oracle.synth(N_IN, 33, T, seed=seed + 30, zero_mod=3)
    lm = oracle.LstmModel()
    lp, la, lc = lm.run(ppm, data)
    ictx_ref, mctx_ref = ictx.copy(), mctx.copy()
    ictx_ref[:, ind_lstm] = np.repeat(lc, 8)
    mctx_ref[:, mix_lstm] = np.repeat(lc, 8)
    io = oracle.IndirectBank(models, *tabs)
    ip, ia = io.run(ictx_ref, bc, bits)
    pred, act = other.copy(), np.zeros((T, N_IN), np.uint8)
    act[:, :8] = act_o[:, :8]
    pred[:, lstm_slot], act[:, lstm_slot] = lp.reshape(-1), la.reshape(-1)
    for i, (a, b_) in enumerate(slots):
        pred[:, a], pred[:, b_] = ip[:, 2 * i], ip[:, 2 * i + 1]
        act[:, a], act[:, b_] = ia[:, 2 * i], ia[:, 2 * i + 1]
    mo = oracle.Bank(N_IN, topo.skip, topo.mixers)
    p_ref, o_ref = mo.run(pred, act, mctx_ref, bits)
    act_host = np.zeros((T, N_IN), np.uint8)
    act_host[:, :8] = act_o[:, :8]
    act_host[:, lstm_slot] = 0
    return dict(T=T, ppm=ppm, bits=bits, bc=bc, ictx=ictx, mctx=mctx, other=other, act=act_host, p=p_ref, o=o_ref,
                mo=mo, io=io, lm=lm, pred_all=pred, act_all=act, ictx_ref=ictx_ref, mctx_ref=mctx_ref)


@pytest.mark.parametrize("path", ["step", "commit", "device_fetch"])
def test_chain_in_lock_step_equals_oracle_chain(gpu, oracle, path, monkeypatch):
    """Three streams of 130 / 101 / 7 bytes (an LSTM backward pass at byte 100 inside the first two; the third ends
    early and sits the remaining steps out; each stream's last step is a learn alone).  Every probability and all 33
    outputs of every bit, and the three banks' state at the end, equal the oracle chain's.  The three ways a step's
    inputs reach the device: gmx_chainstep_step moving them into device memory itself; the caller committing stream by
    stream (what the decoders' threads do) and taking the step in its two halves, launch and wait; and -- as on a host
    without a large BAR -- the step's first kernel fetching them from pinned memory."""
    if path == "device_fetch":
        monkeypatch.setenv("GMX_CS_NO_BAR", "1")
    _, z = goldenlib.load("ind_stock41")
    tabs = (z["ns_next"], z["rm_next"])
    models = topology.stock_indirect()
    topo = topology.stock(90)
    K, S = len(models), 3
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    lg, ig, mg = gpu.LstmGroup(S), gpu.IndirectGroup(models, *tabs, S, slots=slots), gpu.MixerGroup(topo, S)
    rng = np.random.default_rng(5)
    st = [chain_stream(oracle, models, tabs, topo, slots, nb, 40 + s, rng) for s, nb in enumerate((130, 101, 7))]
    for s in range(S):
        lg.set_weights(oracle.LstmModel().weights(), stream=s)
    cs = gpu.ChainStep(mg, ig, lg, lstm_slot=1, mixer_ctx_col=22, ind_ctx_col=16)
    for t in range(max(x["T"] for x in st) + 1):
        for s, x in enumerate(st):
            w = 0
            if 0 < t <= x["T"]:
                w |= LEARN
                cs.bits[s] = x["bits"][t - 1]
            if t < x["T"]:
                w |= PREDICT
                cs.predictions[s, :90] = x["other"][t]
                cs.set_active(s, x["act"][t])
                cs.contexts[s] = x["mctx"][t]
                cs.ind_contexts[s] = x["ictx"][t]
                cs.bit_contexts[s] = x["bc"][t]
                if t % 8 == 0:
                    cs.ppm[s] = x["ppm"][t // 8]
            cs.what[s] = w
            if path == "commit" and w:
                cs.commit(s)
        if path == "commit":
            cs.launch()
            if t == 3:
                with pytest.raises(gpu.GmxError):   # nothing of the object is committed while its step is under way
                    cs.commit(0)
            cs.wait()
        else:
            cs.step()
        for s, x in enumerate(st):
            if t < x["T"]:
                assert cs.p[s].view(np.uint32) == x["p"][t].view(np.uint32), (s, t)
                assert np.array_equal(u32(cs.outputs[s]), u32(x["o"][t])), (s, t)
    cs.close()
    for s, x in enumerate(st):
        assert mg.export(s) == (x["mo"].export_long(), x["mo"].export_short()), s
        assert ig.export(s) == x["io"].export(), s
        w, o = lg.get_weights(s)
        assert np.array_equal(u32(w), u32(x["lm"].weights())) and np.array_equal(u32(o), u32(x["lm"].output_layer())), s
    for x in (lg, ig, mg):
        x.close()


@pytest.mark.parametrize("family,device_fetch", [("indirect", False), ("lstm", False), ("indirect", True), ("lstm", True)])
def test_one_model_family_on_the_device(gpu, oracle, family, device_fetch, monkeypatch):
    """gmx_chainstep with the Indirect models but no LSTM, and with the LSTM but no Indirect models: what the absent family
    would have produced -- predictions, active bits, lstm_prediction_context in the context columns that read it --
    comes in the caller's records like any host-side model's (the oracle chain's own values), and the step's launches
    are the family's kernel on its own (gmx_indirect_step_kernel<false> / gmx_lstm_bitstep_kernel) in front of the
    mixers'.  Two streams of 27 and 10 bytes; every probability and output, and the banks at the end.  device_fetch: the
    step's inputs fetched by a launch of their own at its head (gmx_step_upload_kernel), as on a host without a large BAR."""
    if device_fetch:
        monkeypatch.setenv("GMX_CS_NO_BAR", "1")
    _, z = goldenlib.load("ind_stock41")
    tabs = (z["ns_next"], z["rm_next"])
    models = topology.stock_indirect()
    topo = topology.stock(90)
    K, S = len(models), 2
    slots = [(8 + 2 * i, 9 + 2 * i) for i in range(K)]
    rng = np.random.default_rng(6)
    st = [chain_stream(oracle, models, tabs, topo, slots, nb, 50 + s, rng) for s, nb in enumerate((27, 10))]
    mg = gpu.MixerGroup(topo, S)
    ig = gpu.IndirectGroup(models, *tabs, S, slots=slots) if family == "indirect" else None
    lg = gpu.LstmGroup(S) if family == "lstm" else None
    if lg:
        for s in range(S):
            lg.set_weights(oracle.LstmModel().weights(), stream=s)
    cs = gpu.ChainStep(mg, ig, lg, lstm_slot=1 if lg else -1, mixer_ctx_col=22 if lg else -1)
    dev_slots = [i for ab in slots for i in ab] if ig else [1]   # what the device fills in itself
    for t in range(max(x["T"] for x in st) + 1):
        for s, x in enumerate(st):
            w = 0
            if 0 < t <= x["T"]:
                w |= LEARN
                cs.bits[s] = x["bits"][t - 1]
            if t < x["T"]:
                w |= PREDICT
                cs.predictions[s, :90] = x["pred_all"][t]
                act = x["act_all"][t].copy()
                act[dev_slots] = 0
                cs.set_active(s, act)
                cs.contexts[s] = x["mctx_ref"][t] if ig else x["mctx"][t]
                if ig:
                    cs.ind_contexts[s] = x["ictx_ref"][t]
                    cs.bit_contexts[s] = x["bc"][t]
                if lg and t % 8 == 0:
                    cs.ppm[s] = x["ppm"][t // 8]
            cs.what[s] = w
        cs.step()
        for s, x in enumerate(st):
            if t < x["T"]:
                assert cs.p[s].view(np.uint32) == x["p"][t].view(np.uint32), (s, t)
                assert np.array_equal(u32(cs.outputs[s]), u32(x["o"][t])), (s, t)
    cs.close()
    for s, x in enumerate(st):
        assert mg.export(s) == (x["mo"].export_long(), x["mo"].export_short()), s
        if ig:
            assert ig.export(s) == x["io"].export(), s
        if lg:
            w, o = lg.get_weights(s)
            assert np.array_equal(u32(w), u32(x["lm"].weights())) and np.array_equal(u32(o), u32(x["lm"].output_layer())), s
    for x in (lg, ig, mg):
        if x:
            x.close()


@pytest.mark.parametrize("kind", ["stock", "synth3", "stock_device_fetch"])
def test_mixers_alone_in_lock_step(gpu, oracle, kind, monkeypatch):
    """gmx_chainstep without Indirect models and LSTM: the caller's records carry all inputs (the mixers-only drop-in).
    The reference's shape through gmx_stock_step_kernel (its inputs stored by the host, or fetched by the upload launch),
    the 256-input 24/8/1 bank through a learn and a forward launch of the general kernel."""
    if kind == "stock_device_fetch":
        monkeypatch.setenv("GMX_CS_NO_BAR", "1")
        kind = "stock"
    topo = topology.stock(90) if kind == "stock" else topology.synth3(256, table0=1 << 8)
    N, M, S, T = topo.n_inputs, topo.n_mixers, 4, 300
    mg = gpu.MixerGroup(topo, S)
    cs = gpu.ChainStep(mg)
    recs = [oracle.synth(N, M, T, seed=9 + s, ctx_mode=3, ctx_mod=5, zero_mod=4, bit_mode=1) for s in range(S)]
    refs = []
    for pred, act, ctx, bits in recs:
        ob = oracle.Bank(N, topo.skip, topo.mixers)
        refs.append((ob,) + ob.run(pred, act, ctx, bits))
    ends = [T, T - 37, 5, T]
    for t in range(T + 1):
        for s, (pred, act, ctx, bits) in enumerate(recs):
            w = (LEARN if 0 < t <= ends[s] else 0) | (PREDICT if t < ends[s] else 0)
            if w & LEARN:
                cs.bits[s] = bits[t - 1]
            if w & PREDICT:
                cs.predictions[s, :N] = pred[t]
                cs.set_active(s, act[t])
                cs.contexts[s] = ctx[t]
            cs.what[s] = w
        cs.step()
        for s in range(S):
            if t < ends[s]:
                assert cs.p[s].view(np.uint32) == refs[s][1][t].view(np.uint32), (s, t)
                assert np.array_equal(u32(cs.outputs[s]), u32(refs[s][2][t])), (s, t)
    cs.close()
    for s in (0, 3):
        assert mg.export(s) == (refs[s][0].export_long(), refs[s][0].export_short())
    mg.close()


def test_step_protocol_errors(gpu):
    topo = topology.stock(90)
    mg = gpu.MixerGroup(topo, 2)
    cs = gpu.ChainStep(mg)
    cs.what[:] = [LEARN, 0]
    with pytest.raises(gpu.GmxError):   # a learn without its forward
        cs.step()
    cs.what[:] = [PREDICT, PREDICT]
    cs.step()
    with pytest.raises(gpu.GmxError):   # a second Predict before the first was learned from
        cs.step()
    cs.what[:] = [LEARN | PREDICT, 0]
    with pytest.raises(gpu.GmxError):   # stream 1 sits out between its Predict and its Learn
        cs.step()
    cs.what[:] = [LEARN | PREDICT, LEARN]
    cs.bits[:] = [1, 0]
    cs.step()
    cs.what[:] = [0, 0]
    cs.step()                            # nobody asks: nothing happens
    cs.what[:] = [LEARN, 0]
    cs.step()
    # the step in two halves: a second launch before the wait is refused, and so is a commit; waiting twice is not
    cs.what[:] = [PREDICT, PREDICT]
    cs.launch()
    with pytest.raises(gpu.GmxError):
        cs.launch()
    with pytest.raises(gpu.GmxError):
        cs.commit(1)
    cs.wait()
    cs.wait()
    with pytest.raises(gpu.GmxError):   # no such stream
        cs.commit(2)
    cs.what[:] = [LEARN, LEARN]
    cs.bits[:] = [0, 1]
    cs.commit(0)
    cs.step()
    cs.close()
    mg.close()
