"""The LSTM byte-model banks (gmx_lstm.hip behind gmx_lstm_* of include/gmxmix.h) against the
oracle restatement, which tests/test_oracle_lstm.py pins to the real reference LstmModel: every
bit prediction and active flag, lstm_prediction_context, and the learned weights after backward
passes -- bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def u32(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def run_gpu(gpu, g, streams, chunk, learn=True):
    S, N = len(streams), len(streams[0][1])
    b = gpu.LstmBatch(g, chunk)
    P = np.zeros((S, N, 8), np.float32)
    A = np.zeros((S, N, 8), np.uint8)
    Cx = np.zeros((S, N), np.uint32)
    for n0 in range(0, N, chunk):
        n = min(chunk, N - n0)
        for s, (ppm, data) in enumerate(streams):
            b.ppm[s, :n] = ppm[n0:n0 + n]
            b.bytes[s, :n] = data[n0:n0 + n]
        b.upload(n)
        g.run(b, n, learn=learn)
        b.download(n)
        b.wait()
        P[:, n0:n0 + n] = b.predictions[:, :n]
        A[:, n0:n0 + n] = b.active[:, :n]
        Cx[:, n0:n0 + n] = b.contexts[:, :n]
    b.close()
    return P, A, Cx


@pytest.fixture(params=["1", "2"], ids=["one_workgroup_per_cu_build", "two_per_cu_build"])
def lstm_build(request):
    """The batched LSTM kernel has a build for launches of at most one workgroup per CU (<= 256 streams: every launch
    of these tests) and one for more streams (what bench.py's 1 024 streams run): GMX_LSTM_BUILD, read by the launcher
    at every launch, puts each test through both."""
    import os
    old = os.environ.get("GMX_LSTM_BUILD")
    os.environ["GMX_LSTM_BUILD"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("GMX_LSTM_BUILD", None)
    else:
        os.environ["GMX_LSTM_BUILD"] = old


def test_lstm_kernel_equals_oracle_through_backward_passes(gpu, oracle, lstm_build):
    S, N = 3, 450                       # four backward passes + Adam steps, launches that split them
    refs, streams = [], []
    for s in range(S):
        m = oracle.LstmModel()          # srand(0xDEADBEEF) + the reference's constructor chain
        ppm, data = oracle.lstm_synth(N, seed=11 + s, mask=15 if s == 1 else 255)
        refs.append((m, m.weights()) + m.run(ppm, data))
        streams.append((ppm, data))
    g = gpu.LstmGroup(S)
    for s in range(S):
        g.set_weights(refs[s][1], stream=s)
    P, A, Cx = run_gpu(gpu, g, streams, chunk=170)
    for s in range(S):
        m, _, pred, act, ctx = refs[s]
        bad = np.argwhere(u32(P[s]) != u32(pred))
        assert len(bad) == 0, (s, bad[:4], P[s][tuple(bad[0])], pred[tuple(bad[0])])
        assert np.array_equal(A[s], act) and np.array_equal(Cx[s], ctx)
        w, o = g.get_weights(s)
        assert np.array_equal(u32(w), u32(m.weights())), s
        assert np.array_equal(u32(o), u32(m.output_layer())), s
    g.close()


def test_lstm_generation_mode_and_restart(gpu, oracle):
    """Predict without Learn (runner-utils.cpp:199-209) changes nothing but the recurrent state;
    a second group fed the same weights reproduces the first."""
    N = 230
    m = oracle.LstmModel()
    w0 = m.weights()
    ppm, data = oracle.lstm_synth(N, seed=5, mask=63)
    p1, a1, c1 = m.run(ppm[:130], data[:130])
    p2, a2, c2 = m.run(ppm[130:], data[130:], learn=False)
    g = gpu.LstmGroup(1)
    g.set_weights(w0)
    P, A, Cx = run_gpu(gpu, g, [(ppm[:130], data[:130])], chunk=130)
    P2, A2, Cx2 = run_gpu(gpu, g, [(ppm[130:], data[130:])], chunk=100, learn=False)
    assert np.array_equal(u32(P[0]), u32(p1)) and np.array_equal(u32(P2[0]), u32(p2))
    assert np.array_equal(A2[0], a2) and np.array_equal(Cx2[0], c2)
    w, o = g.get_weights(0)
    assert np.array_equal(u32(w), u32(m.weights())) and np.array_equal(u32(o), u32(m.output_layer()))
    g.close()


def test_lstm_generation_matches_reference_golden(gpu, oracle):
    """Generation (runner-utils.cpp:199-209: Predict and Perceive, never Learn) against the REAL reference's
    LstmModel (tests/golden/lstm_generation.npz): 230 bytes learned through the batched kernel, then 130 bytes
    through gmx_lstm_forward alone -- Lstm::Predict on output layers no Perceive refreshes, across an epoch wrap --
    the bits coded from the byte distribution on the host as the adapter does; every bit prediction, the
    checksum, and the file the reference wrote at the end, in the state of its TestGeneration checkpoints
    (tester.cpp:312): the newest forward never perceived."""
    import goldenlib
    meta, z = goldenlib.load("lstm_generation")
    kw = meta["synth"]
    N, N0 = meta["bytes"], kw["nolearn_from"]
    ppm, data = oracle.lstm_synth(N, seed=kw["seed"], mask=kw["mask"])
    m = oracle.LstmModel()                      # (only its bit-from-probs helper and initial weights are used)
    g = gpu.LstmGroup(1)
    g.set_weights(m.weights())
    P, A, Cx = run_gpu(gpu, g, [(ppm[:N0], data[:N0])], chunk=N0)
    assert np.array_equal(u32(P[0]), z["pred"][:N0]) and np.array_equal(A[0], z["active"][:N0])
    import ctypes
    m._lb, m._pr, m._cx = ctypes.c_uint32(0), ctypes.c_float(float(P[0, -1, -1])), ctypes.c_uint32(0)  # the slot as the last learned bit left it
    last = int(data[N0 - 1])
    tmb = None
    for n in range(N0, N):
        probs, ctx = g.forward(ppm[n], last)
        pr, act, tmb = m.bits_from_probs(probs, data[n])
        assert np.array_equal(u32(pr), z["pred"][n]) and np.array_equal(act, z["active"][n]) and ctx == z["ctx"][n], n
        last = int(data[n])
    lng, sh = g.export(0)                       # between that forward and a Perceive that never comes
    sh = bytearray(sh)
    for i, v in enumerate(tmb):                 # the range is the host's (include/gmxmix.h)
        sh[4 * i:4 * i + 4] = int(v).to_bytes(4, "little")
    assert (meta["top"], meta["mid"], meta["bot"]) == tmb
    assert len(sh) == meta["short_size"] and oracle.fnv64_bytes(bytes(sh)) == meta["short_hash"]
    assert oracle.fnv64_bytes(lng) == meta["long_hash"]
    g.close()


@pytest.mark.parametrize("name", ["lstm_short", "lstm_alphabet16", "lstm_long", "lstm_update_limit"])
def test_lstm_kernel_matches_reference_goldens(gpu, oracle, name, lstm_build):
    """The fixtures made from the REAL reference LstmModel (tests/golden/lstm_*.npz), replayed
    through the HIP path from the seed: dumped bit predictions, the checksum over every bit, the
    hash of the learned weights and output layer."""
    import goldenlib
    meta, z = goldenlib.load(name)
    kw = meta["synth"]
    N = meta["bytes"]
    ppm, data = oracle.lstm_synth(N, seed=kw.get("seed", 0), mask=kw.get("mask", 255))
    m = oracle.LstmModel()
    assert m.weights_hash() == meta["init_weights_hash"]
    g = gpu.LstmGroup(1)
    g.set_weights(m.weights())
    P, A, Cx = run_gpu(gpu, g, [(ppm, data)], chunk=min(N, 20000))   # lstm_update_limit: 3050 backward passes
    D = meta["dump"]
    if D:
        assert np.array_equal(u32(P[0, :D]), z["pred"]) and np.array_equal(A[0, :D], z["active"])
        assert np.array_equal(Cx[0, :D], z["ctx"])
    # the harness' running checksum: per bit (prediction, active), per byte the context after bit 0
    rec = np.zeros((N, 8 * 5 + 4), np.uint8)
    for k in range(8):
        off = 5 * k + (4 if k > 0 else 0)
        rec[:, off:off + 4] = P[0, :, k].copy().view(np.uint8).reshape(N, 4)
        rec[:, off + 4] = A[0, :, k]
        if k == 0:
            rec[:, 5:9] = Cx[0].copy().view(np.uint8).reshape(N, 4)
    assert oracle.fnv64_bytes(rec.reshape(-1)) == meta["h64"]
    w, o = g.get_weights(0)
    assert oracle.fnv64_bytes(np.concatenate([o.reshape(-1).view(np.uint8), w.reshape(-1).view(np.uint8)])) == meta["long_hash"]
    # the checkpoint the real LstmModel wrote at this point: size and hash of its .short stretch
    # (every field, the scratch a backward pass leaves behind included), hash of its .long section
    lng, sh = g.export(0)
    assert len(sh) == meta["short_size"] and oracle.fnv64_bytes(sh) == meta["short_hash"]
    assert oracle.fnv64_bytes(lng) == meta["long_hash"]
    assert g.memory_usage() == meta["usage"]
    g.close()


def test_lstm_checkpoint_import_export_copy(gpu, oracle):
    """gmx_lstm_export / import / copy against the oracle's LstmModel::WriteToDisk / ReadFromDisk
    (pinned to the reference's bytes by tests/test_oracle_lstm.py): same files at every stage, and a
    bank restored from a file continues exactly like the model that wrote it."""
    N, cut = 390, 230                              # two backward passes, 30 bytes into the third epoch
    ppm, data = oracle.lstm_synth(N, seed=9, mask=127)
    m = oracle.LstmModel()
    g = gpu.LstmGroup(3)
    g.set_weights(m.weights(), stream=2)
    lng, sh = g.export(2)                          # a model nobody has used yet
    assert lng == m.export_long() and sh == m.export_short()
    m.run(ppm[:cut], data[:cut])
    g.import_(m.export_long(), m.export_short(), stream=1)       # the oracle's file into a bank ...
    lng, sh = g.export(1)
    assert lng == m.export_long() and sh == m.export_short()     # ... and out again, byte for byte
    g2 = gpu.LstmGroup(1)
    g2.copy_from(g, src_stream=1)
    assert g2.export(0) == (lng, sh)
    pred, act, ctx = m.run(ppm[cut:], data[cut:])                # on through the third backward pass
    for grp, st in ((g, 1), (g2, 0)):
        b = gpu.LstmBatch(grp, N - cut)
        b.ppm[st, :] = ppm[cut:]
        b.bytes[st, :] = data[cut:]
        for other in range(grp.S):
            if other != st:
                b.ppm[other, :] = 1.0 / 256
                b.bytes[other, :] = 0
        b.upload(N - cut)
        grp.run(b, N - cut, learn=True)
        b.download(N - cut)
        b.wait()
        assert np.array_equal(u32(b.predictions[st]), u32(pred)), st
        assert np.array_equal(b.active[st], act) and np.array_equal(b.contexts[st], ctx)
        b.close()
        assert grp.export(st) == (m.export_long(), m.export_short())
    # the bank written by the device goes back into the oracle and both keep agreeing
    m2 = oracle.LstmModel(srand_seed=3)
    m2.import_state(*g.export(1))
    assert m2.export_short() == m.export_short()
    # Inside a byte -- LstmModel::WriteToDisk works at any bit (lstm-model.cpp:62-68) and the reference's
    # TestGeneration checkpoints after a Predict whose byte is never perceived (tester.cpp:284, :312): the file of a
    # bank between forward and perceive is the reference's there, goes into another bank, and both go on alike
    g.forward(ppm[0], int(data[-1]), stream=1)
    m.predict_byte(ppm[0], int(data[-1]))
    mid = g.export(1)
    assert mid == (m.export_long(), m.export_short())
    g.import_(*mid, stream=0)
    g2.copy_from(g, src_stream=1)                  # ... and a copy taken there
    m.perceive_byte(int(data[0]))
    b0 = int(data[0])
    for grp, st in ((g, 1), (g, 0), (g2, 0)):
        grp.perceive(b0, stream=st)
        lng1, sh1 = grp.export(st)
        # (where a byte has ended the bank writes the range of its eighth bit; the oracle's byte-level calls never
        # coded the bits, its range is still the forward's)
        assert lng1 == m.export_long() and sh1[12:] == m.export_short()[12:], st
        assert [int.from_bytes(sh1[4 * i:4 * i + 4], "little") for i in range(3)] == [b0 | 1, b0 & ~1, b0 & ~1]
    # generation: bytes predicted, never perceived, a checkpoint after each of them
    for n in (1, 2, 3):
        for grp, st in ((g, 1), (g, 0)):
            grp.forward(ppm[n], int(data[n - 1]), stream=st)
        m.predict_byte(ppm[n], int(data[n - 1]))
        assert g.export(1) == g.export(0) == (m.export_long(), m.export_short()), n
    # format errors
    with pytest.raises(gpu.GmxError):
        g.import_(lng[:-4], sh, stream=0)
    bad = bytearray(sh)
    bad[0:4] = (300).to_bytes(4, "little")         # top_ out of range
    with pytest.raises(gpu.GmxError):
        g.import_(lng, bytes(bad), stream=0)
    bad = bytearray(sh)
    bad[8:12] = (201).to_bytes(4, "little")        # bot_ above mid_
    bad[4:8] = (200).to_bytes(4, "little")
    bad[0:4] = (255).to_bytes(4, "little")
    with pytest.raises(gpu.GmxError):
        g.import_(lng, bytes(bad), stream=0)
    g.close()
    g2.close()


@pytest.mark.parametrize("sessions", [1, 0])
def test_lstm_per_byte_surface_for_decoding(gpu, oracle, sessions):
    """gmx_lstm_forward / gmx_lstm_perceive: the byte distribution and lstm_prediction_context one
    byte at a time, through two backward passes -- through the persistent per-byte session (commands in a
    mailbox, the Perceive travelling with the next Predict) and with a kernel launch per call; the session
    also across an idle exit and across calls that make it hand the bank back."""
    import ctypes as C
    import time
    N = 230
    ppm, data = oracle.lstm_synth(N, seed=21, mask=31)
    m = oracle.LstmModel()
    g = gpu.LstmGroup(2)
    g.L.gmx_debug_lstm_use_sessions.argtypes = [C.c_void_p, C.c_int]
    assert g.L.gmx_debug_lstm_use_sessions(g.h, sessions) == 0
    g.set_weights(m.weights(), stream=1)
    last = 0
    with pytest.raises(gpu.GmxError):
        g.perceive(3, stream=1)                    # Perceive before any Predict
    for n in range(215):
        if n == 60:
            time.sleep(0.06)                       # longer than the session's idle timer: its block has left
        if n == 95:
            assert g.memory_usage() > 0 and len(g.export(0)[0]) > 0   # another stream's file: the session hands over
        if n == 140:
            g.sync()
        probs, ctx = g.forward(ppm[n], last, stream=1)
        p_ref, c_ref = m.predict_byte(ppm[n], last)
        assert np.array_equal(u32(probs), u32(p_ref)) and ctx == c_ref, n
        if n == 120:
            # a checkpoint between this byte's Predict and its Perceive (the session hands the bank back with the
            # forward done), restored into the other stream: both perceive the byte and stay the oracle's
            mid = g.export(1)
            assert mid == (m.export_long(), m.export_short())
            g.import_(*mid, stream=0)
            g.perceive(int(data[n]), stream=0)
        if n % 50 == 7:                            # a byte whose bits are predicted but never learned
            last = int(data[n])                    # (generation): the hidden state still moves on
            continue
        g.perceive(int(data[n]), stream=1)
        m.perceive_byte(int(data[n]))
        if n == 120:
            e0, e1 = g.export(0), g.export(1)
            assert e0 == e1 and e1[0] == m.export_long() and e1[1][12:] == m.export_short()[12:]   # (12: the range state)
        last = int(data[n])
    w, o = g.get_weights(1)
    assert np.array_equal(u32(w), u32(m.weights())) and np.array_equal(u32(o), u32(m.output_layer()))
    g.close()
