"""The CPU restatement of the LSTM byte model (oracle/gmx_oracle_lstm.c) against golden vectors
made from the REAL reference LstmModel (tests/golden/lstm_*.npz,
oracle/ref_build/ref_lstm_harness.cpp): the rand()-initialised weights, every bit prediction and
active flag of the dumped bytes, lstm_prediction_context, a checksum over all bits, the learned
weights after all backward passes."""
import numpy as np
import pytest

import goldenlib
from golden.cases import LSTM_CASES


@pytest.mark.parametrize("name", sorted(LSTM_CASES))
def test_lstm_oracle_matches_reference(oracle, name):
    meta, z = goldenlib.load(name)
    m = oracle.LstmModel()                       # srand(0xDEADBEEF) + the constructor chain
    assert m.weights_hash() == meta["init_weights_hash"]
    kw = meta["synth"]
    h, pred, act, ctx = m.run_synth(meta["bytes"], seed=kw.get("seed", 0), mask=kw.get("mask", 255), dump=meta["dump"],
                                    nolearn_from=kw.get("nolearn_from"))
    if meta["dump"]:
        assert np.array_equal(pred.view(np.uint32), z["pred"])
        assert np.array_equal(act, z["active"]) and np.array_equal(ctx, z["ctx"])
    assert h == meta["h64"]
    assert m.weights_hash(with_output_layer=True) == meta["long_hash"]
    # LstmModel::WriteToDisk of the real model at the end of the run: size and FNV-1a of its bytes
    # -- every field, including the scratch ones a backward pass leaves behind
    short = m.export_short()
    assert len(short) == meta["short_size"]
    assert oracle.fnv64_bytes(short) == meta["short_hash"]
    assert oracle.fnv64_bytes(m.export_long()) == meta["long_hash"]


def test_lstm_oracle_checkpoint_round_trip(oracle):
    """ReadFromDisk(WriteToDisk) into a fresh model continues exactly like the original."""
    a = oracle.LstmModel()
    ppm, data = oracle.lstm_synth(380, seed=4, mask=63)
    a.run(ppm[:230], data[:230])
    b = oracle.LstmModel(srand_seed=1)            # different weights until the import
    b.import_state(a.export_long(), a.export_short())
    assert b.export_short() == a.export_short() and b.export_long() == a.export_long()
    b._lb, b._pr, b._cx = type(a._lb)(a._lb.value), type(a._pr)(a._pr.value), type(a._cx)(a._cx.value)
    ra, rb = a.run(ppm[230:], data[230:]), b.run(ppm[230:], data[230:])
    for x, y in zip(ra, rb):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    assert a.export_short() == b.export_short() and a.export_long() == b.export_long()


def test_lstm_learns_the_synthetic_stream(oracle):
    """Sanity of the fixture itself: on the 16-symbol stream the model's bit predictions carry
    information after a few thousand bytes (they are not stuck at the prior)."""
    m = oracle.LstmModel()
    _, pred, act, _ = m.run_synth(3000, seed=7, mask=15, dump=3000)
    late = np.abs(pred[2500:]).mean()
    early = np.abs(pred[:100]).mean()
    assert act[2500:].all() and late > 2 * early
