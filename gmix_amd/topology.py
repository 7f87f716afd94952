"""Mixer topologies: the stock one `Predictor::AddMixers` builds (predictor.cpp:251-358) and
the synthetic ones BASELINE.json / SURVEY.md section 8d name."""
from .bank import Topology

# (gate context, table_size, learning_rate) in construction order -- predictor.cpp:254-325.
STOCK_LAYER0 = [
    ("last_byte", 1 << 8, 0.005), ("recent_bytes[3]", 1 << 8, 0.0055),
    ("second_last_plus_recent", 1 << 16, 0.003), ("last_four_bytes_hash", 1 << 15, 0.0045),
    ("indirect_3_24_1", 1 << 8, 0.006), ("recent_bytes[1]", 1 << 8, 0.004),
    ("longest_match", 1 << 3, 0.0005), ("last_two_bytes_hash", 1 << 16, 0.0035),
    ("recent_bytes[2]", 1 << 8, 0.0065), ("last_three_bytes_hash", 1 << 15, 0.0025),
    ("last_byte", 1 << 8, 0.001), ("last_byte_plus_recent", 1 << 16, 0.002),
    ("interval_16_4", 1 << 4, 0.005), ("interval_16_8", 1 << 8, 0.0045),
    ("interval_16_12", 1 << 12, 0.0055), ("interval_32_3", 1 << 3, 0.004),
    ("interval_32_6", 1 << 6, 0.0035), ("skip_0_2", 1 << 16, 0.006),
    ("interval_32_12", 1 << 12, 0.003), ("interval_64_4", 1 << 4, 0.0065),
    ("interval_64_8", 1 << 8, 0.003), ("interval_64_12", 1 << 12, 0.0025),
    ("lstm_prediction_context", 1 << 8, 0.002), ("always_zero", 1, 0.0005),
]
# predictor.cpp:328-351
STOCK_LAYER1 = [
    ("recent_bytes[1]", 1 << 8, 0.0045), ("always_zero", 1, 0.0035), ("bit_context", 1 << 8, 0.003),
    ("recent_bytes[2]", 1 << 8, 0.002), ("last_byte", 1 << 8, 0.0025), ("bit_context", 1 << 8, 0.00001),
    ("longest_match", 1 << 3, 0.0008), ("always_zero", 1, 0.0004),
]
# predictor.cpp:355-357
STOCK_FINAL = [("always_zero", 1, 0.0005)]


def stock(n_inputs=90, skip=(1,)):
    """The 24/8/1 topology of the reference (90 model predictions, LSTM skip connection)."""
    mixers = ([(0, t, lr) for _, t, lr in STOCK_LAYER0] + [(1, t, lr) for _, t, lr in STOCK_LAYER1] +
              [(2, t, lr) for _, t, lr in STOCK_FINAL])
    return Topology(n_inputs, mixers, skip)


def stock_context_names():
    return [c for c, _, _ in STOCK_LAYER0 + STOCK_LAYER1 + STOCK_FINAL]


def single(n_inputs=256, table_size=1 << 16, lr=0.005):
    """BASELINE.json configs[1]: one layer-0 mixer over n_inputs logits."""
    return Topology(n_inputs, [(0, table_size, lr)], skip=(1,))


def synth3(n_inputs=256, l0=24, l1=8, table0=1 << 16, table1=1 << 8):
    """SURVEY.md Appendix A.3's 24/8/1 synthetic bank."""
    mixers = [(0, table0, 0.005)] * l0 + [(1, table1, 0.003)] * l1 + [(2, 1, 0.0005)]
    return Topology(n_inputs, mixers, skip=(1,))


# The 41 Indirect models of the reference in construction order (predictor.cpp:78-120 AddIndirect,
# :122-185 AddSkip, :210-250 AddDoubleIndirect): (context variable, table_size, learning_rate).
# Each owns 256*table_size+1 one-byte states twice over and produces two of the mixers' 90 inputs.
STOCK_INDIRECT = (
    [("last_byte", 1 << 8, 0.02), ("last_two_bytes_hash", 1 << 16, 0.02),
     ("last_three_bytes_hash", 1 << 15, 0.02), ("last_three_bytes_hash", 1 << 16, 0.02),
     ("last_four_bytes_hash", 1 << 15, 0.02), ("last_five_bytes_hash", 1 << 15, 0.02),
     ("last_six_bytes_hash", 1 << 15, 0.02)] +
    [(f"recent_bytes[{i}]", 1 << 8, 0.02) for i in range(1, 10)] +
    [("lstm_prediction_context", 1 << 8, 0.02)] +
    [(c, 1 << 16, 0.02) for c in ("skip_1_2", "skip_1_2_3", "skip_0_2", "skip_0_2_3", "skip_1_2_3_4", "skip_0_3",
                                  "skip_0_4", "skip_0_5", "skip_0_2_3_4", "skip_0_3_4", "skip_0_6", "skip_0_7",
                                  "skip_0_1_3_4", "skip_0_4_5", "skip_0_1_2_4")] +
    [("indirect_1_8_1", 1 << 8, 1.0 / 200), ("indirect_1_8_2", 1 << 16, 1.0 / 200),
     ("indirect_1_8_3", 1 << 15, 1.0 / 200), ("indirect_2_16_1", 1 << 8, 1.0 / 200),
     ("indirect_2_16_2", 1 << 16, 1.0 / 200), ("indirect_2_16_3", 1 << 15, 1.0 / 200),
     ("indirect_3_24_1", 1 << 8, 1.0 / 200), ("indirect_4_24_2", 1 << 16, 1.0 / 200),
     ("indirect_4_24_3", 1 << 15, 1.0 / 200)])


def stock_indirect():
    """[(table_size, learning_rate)] of the 41 stock Indirect models (learning rates as the float
    the reference's constructor receives)."""
    import numpy as np
    return [(t, float(np.float32(lr))) for _, t, lr in STOCK_INDIRECT]
