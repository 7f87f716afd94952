"""The CPU restatement of the Indirect models (oracle/gmx_oracle_ind.c) against golden vectors made
from the REAL reference class (tests/golden/ind_*.npz, oracle/ref_build/ref_indirect_harness.cpp):
every prediction slot and active flag of the dumped bits, a checksum over all bits, the
indirect section of the reference's .long checkpoint, GetMemoryUsage."""
import numpy as np
import pytest

import goldenlib
from golden.cases import IND_CASES


def replay(oracle, name):
    meta, models, ctx, bc, bits, nolearn, z = goldenlib.ind_case(name)
    b = oracle.IndirectBank(models, z["ns_next"], z["rm_next"])
    pred, act = b.run(ctx, bc, bits, nolearn_from=nolearn)
    return meta, z, b, pred, act


@pytest.mark.parametrize("name", sorted(IND_CASES))
def test_indirect_oracle_matches_reference(oracle, name):
    meta, z, b, pred, act = replay(oracle, name)
    D, K = meta["dump"], len(meta["models"])
    if D:
        assert np.array_equal(pred[:D].view(np.uint32), z["pred"])
        assert np.array_equal(act[:D], np.unpackbits(z["active"], axis=1, bitorder="little")[:, :2 * K])
    assert oracle.ind_fnv64(pred, act) == meta["h64"]
    assert [b.memory_usage(i) for i in range(K)] == meta["usage"]
    e = b.export()
    assert len(e) == meta["long_len"] and goldenlib.sha256(e) == meta["long_sha256"]
    if len(z["long"]):
        assert e == z["long"].tobytes()


def test_state_machine_tables_are_the_references(oracle):
    """All fixtures carry the same two 256x2 next-state tables, dumped from the reference's
    ShortTermMemory::nonstationary / ::run_map; the run map's rule is simple enough to restate
    (contexts/run-map.cpp:3-19) and check."""
    tabs = [(goldenlib.load(n)[1]["ns_next"], goldenlib.load(n)[1]["rm_next"]) for n in sorted(IND_CASES)]
    for ns, rm in tabs[1:]:
        assert np.array_equal(ns, tabs[0][0]) and np.array_equal(rm, tabs[0][1])
    rm = tabs[0][1].reshape(256, 2)
    for s in range(256):
        zero = s + 1 if s < 127 else (1 if s >= 128 else s)
        one = 128 if s < 128 else (s + 1 if s < 255 else s)
        assert (rm[s, 0], rm[s, 1]) == (zero, one), s
    assert tabs[0][0].max() <= 254 or 255 in tabs[0][0]  # states are bytes; 255 marks "never seen"


def test_indirect_edge_semantics(oracle):
    """Never-seen contexts leave the slots alone (stale value, not active); a zero logit is
    stored but not active (short-term-memory.cpp:193-197); index arithmetic wraps in 32 bits."""
    _, z = goldenlib.load("ind_tiny_dense")
    b = oracle.IndirectBank([(1, 0.5), (3, 0.25)], z["ns_next"], z["rm_next"])
    p, a = b.predict([0, 0], 0)
    assert not a.any() and not p.any()                      # nothing seen yet
    b2 = oracle.IndirectBank([(1, 0.5)], z["ns_next"], z["rm_next"])
    for bit in (1, 0, 1):                                    # run map of this row: 0 -> 128 -> 1 -> 128
        p, a = b2.predict([7], 0)
        assert a[1] == 0 and p[1] == 0                      # states exist, their predictions are still 0
        b2.learn(bit)
    p, a = b2.predict([7], 0)                                # run-map state 128 again: trained once
    assert a[1] == 1 and p[1] == np.float32(-0.25)          # (0 - Logistic(0)) * 0.5 when it left 128
    p2, a2 = b2.predict([0x01000007], 0)                     # (ctx << 8) drops the top byte: same row
    assert a2[1] == 1 and p2[1] == p[1]
    p3, a3 = b2.predict([8], 0)                              # never seen: slot keeps its stale value
    assert a3[1] == 0 and p3[1] == p[1]
