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
    h, pred, act, ctx = m.run_synth(meta["bytes"], seed=kw.get("seed", 0), mask=kw.get("mask", 255), dump=meta["dump"])
    if meta["dump"]:
        assert np.array_equal(pred.view(np.uint32), z["pred"])
        assert np.array_equal(act, z["active"]) and np.array_equal(ctx, z["ctx"])
    assert h == meta["h64"]
    assert m.weights_hash(with_output_layer=True) == meta["long_hash"]


def test_lstm_learns_the_synthetic_stream(oracle):
    """Sanity of the fixture itself: on the 16-symbol stream the model's bit predictions carry
    information after a few thousand bytes (they are not stuck at the prior)."""
    m = oracle.LstmModel()
    _, pred, act, _ = m.run_synth(3000, seed=7, mask=15, dump=3000)
    late = np.abs(pred[2500:]).mean()
    early = np.abs(pred[:100]).mean()
    assert act[2500:].all() and late > 2 * early
