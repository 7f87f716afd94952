"""Synthetic inputs for the auxiliary benches (numpy, host side, outside every timed region): records and initial
weights of the right SHAPE and range for the LSTM byte model.  Nothing here is checked against anything -- the parity
tests take their streams from oracle/ (the checker); a bench only needs plausible numbers."""
import numpy as np


def lstm_records(n_bytes, seed=1, mask=63):
    """(ppm[n_bytes][256] float32: a byte distribution per byte, peaked on a few symbols like PPMd's; bytes[n_bytes])."""
    rng = np.random.default_rng(seed)
    data = (rng.integers(0, 256, n_bytes) & mask).astype(np.uint8)
    ppm = rng.random((n_bytes, 256), dtype=np.float32) * np.float32(0.002)
    ppm[np.arange(n_bytes), data] += np.float32(0.6)                      # the coded byte is likely ...
    ppm[np.arange(n_bytes), (data.astype(np.int64) * 7 + 3) & 255] += np.float32(0.2)   # ... a rival a bit less
    ppm /= ppm.sum(axis=1, keepdims=True)
    return np.ascontiguousarray(ppm, np.float32), data


def lstm_initial_weights(seed=0xDEADBEEF):
    """[3][50][563] gate weights drawn like LstmLayer's constructor draws them (lstm-layer.cpp:179-194): uniform in
    +-sqrt(6 / 512), the forget gate's last column 1 -- from numpy's generator, not rand()."""
    rng = np.random.default_rng(seed)
    val = np.float32(np.sqrt(np.float32(6.0) / np.float32(512)))
    w = (rng.random((3, 50, 563), dtype=np.float32) * (2 * val) - val).astype(np.float32)
    w[0, :, 562] = 1.0
    return w
