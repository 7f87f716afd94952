"""Scenarios pinned by golden vectors.  `make_golden.py` runs each through the REFERENCE's own
Mixer (oracle/_ref/ref_mixer_harness) in the build container; the tests replay them through
the oracle (CPU) and through the HIP path (GPU) from the seed alone."""
from gmix_amd import topology


def _t(kind, *a, **k):
    return getattr(topology, kind)(*a, **k)


# name -> (topology factory, T bits, bits fully dumped, synthetic-stream kwargs)
CASES = {
    # SURVEY.md Appendix A.3 runs (hash-only, long):
    "a3_single256": (lambda: _t("single", 256, 1 << 16), 2_000_000, 0, {}),
    "a3_synth3_n256": (lambda: _t("synth3", 256), 200_000, 0, {}),
    "a3_synth3_n90": (lambda: _t("synth3", 90), 400_000, 0, {}),
    # short, fully dumped runs covering the reference's edge cases:
    "single256": (lambda: _t("single", 256, 1 << 16), 3000, 3000, {}),
    "single256_rowrepeat": (lambda: _t("single", 256, 1 << 16), 3000, 3000, dict(ctx_mode=1, ctx_mod=3)),
    "single90_odd": (lambda: _t("single", 90, 1000), 2000, 2000, dict(ctx_mode=3, ctx_mod=7, zero_mod=5)),
    "synth3_n90_sticky_silent": (lambda: _t("synth3", 90, table0=1 << 12), 2500, 2500,
                                 dict(ctx_mode=3, ctx_mod=5, zero_mod=7)),
    "synth3_n256": (lambda: _t("synth3", 256, table0=1 << 10), 1500, 1500, dict(ctx_mode=1, ctx_mod=50)),
    "stock90_learnable": (lambda: _t("stock", 90), 4000, 4000, dict(ctx_mode=2, zero_mod=9, bit_mode=1)),
    "stock90_smallctx": (lambda: _t("stock", 90), 3000, 3000, dict(ctx_mode=1, ctx_mod=2, bit_mode=1)),
    "stock90_shrink": (lambda: _t("stock", 90), 2600, 600, dict(ctx_mode=1, ctx_mod=2, bit_mode=1, seed=4242)),
    "tiny_two_skips": (lambda: topology.Topology(7, [(0, 5, .02), (0, 1, .01), (0, 300, .03), (1, 2, .01),
                                                     (1, 9, .02), (2, 1, .005)], skip=(0, 3)), 3000, 3000,
                       dict(ctx_mode=1, ctx_mod=11, zero_mod=3, bit_mode=1)),
    "no_final_no_skip": (lambda: topology.Topology(33, [(0, 64, .01)] * 5 + [(1, 8, .01)] * 2, skip=()),
                         2000, 2000, dict(ctx_mode=1, ctx_mod=70, bit_mode=1)),
    # generation mode: Learn stops at bit 800 (runner-utils.cpp:199-209)
    "stock90_nolearn_tail": (lambda: _t("stock", 90), 1200, 1200,
                             dict(ctx_mode=3, ctx_mod=9, bit_mode=1, nolearn_from=800)),
}


# Indirect models, through oracle/_ref/ref_indirect_harness:
# name -> (models [(table_size, lr)], T bits, bits fully dumped, kwargs of gmx_ind_synth.h)
IND_CASES = {
    # the 41 stock models; contexts from small, medium and 32-bit domains (hash wrap, collisions)
    "ind_stock41": (topology.stock_indirect, 12000, 1500, dict(ctx_mod=(300, 0, 70000, 5))),
    # tables so small that they fill up: the dense branch of the checkpoint format, index wrap
    "ind_tiny_dense": (lambda: [(1, 0.02), (2, 0.005), (3, 0.1), (1, 0.5)], 30000, 2500,
                       dict(ctx_mod=(3, 2, 5, 1), seed=77)),
    # generation mode: Learn stops (runner-utils.cpp:199-209)
    "ind_nolearn_tail": (lambda: [(256, 0.02), (4096, 0.005), (65536, 0.02)], 4000, 4000,
                         dict(ctx_mod=(40, 900, 0, 40), seed=5, nolearn_from=3000)),
    # long run, checksum only
    "ind_long": (lambda: [(256, 0.02), (65536, 0.02), (32768, 0.005), (256, 0.005), (65536, 0.02), (1, 0.02)],
                 1_000_000, 0, dict(ctx_mod=(50, 0, 3000, 7), seed=99)),
}


# LSTM byte model, through oracle/_ref/ref_lstm_harness:
# name -> (bytes, bytes fully dumped, kwargs of gmx_lstm_synth.h)
LSTM_CASES = {
    "lstm_short": (300, 300, dict()),                       # three backward passes, every bit dumped
    "lstm_alphabet16": (5000, 100, dict(seed=7, mask=15)),  # a learnable 16-symbol stream
    "lstm_long": (20000, 0, dict(seed=99)),                 # 200 backward passes, checksums only
    "lstm_update_limit": (305000, 0, dict(seed=3, mask=63)),  # Adam's step count reaches update_limit_ = 3000
    # generation (runner-utils.cpp:199-209): from byte 230 on Predict and Perceive only -- 130 forwards on output
    # layers no Perceive refreshes, across the epoch wrap at byte 300; the file at the end is the kind the
    # reference's TestGeneration writes (tester.cpp:312): the newest forward never perceived
    "lstm_generation": (360, 360, dict(seed=11, mask=63, nolearn_from=230)),
}
