"""Host-side handles of the LSTM byte-model banks (gmx_lstm / gmx_lstm_batch of include/gmxmix.h):
test and bench harness."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GmxError, check

NC, W, H, NO, HID = 50, 563, 100, 256, 51


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class LstmGroup:
    def __init__(self, n_streams=1, device=0):
        self.L = _lib.lib()
        self.S = int(n_streams)
        h = C.c_void_p()
        check(self.L.gmx_lstm_create(C.byref(h), self.S, device), "gmx_lstm_create")
        self.h = h

    def set_cu_mask(self, words=None):
        """Compute units this bank's kernels may use: 32-bit words, bit i = CU i (None: all)."""
        w = list(words) if words else []
        arr = (C.c_uint32 * max(1, len(w)))(*w)
        check(self.L.gmx_lstm_set_cu_mask(self.h, arr, len(w)), "gmx_lstm_set_cu_mask")

    def close(self):
        if getattr(self, "h", None):
            self.L.gmx_lstm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def bank_bytes(self):
        return self.L.gmx_lstm_bank_bytes(self.h)

    def reset(self):
        check(self.L.gmx_lstm_reset(self.h), "gmx_lstm_reset")

    def sync(self):
        check(self.L.gmx_lstm_sync(self.h), "gmx_lstm_sync")

    def set_weights(self, w, stream=0):
        w = np.ascontiguousarray(w, np.float32)
        assert w.shape == (3, NC, W)
        check(self.L.gmx_lstm_set_weights(self.h, stream, _vp(w)), "gmx_lstm_set_weights")

    def get_weights(self, stream=0, output_layer=True):
        w = np.zeros((3, NC, W), np.float32)
        o = np.zeros((H, NO, HID), np.float32) if output_layer else None
        check(self.L.gmx_lstm_get_weights(self.h, stream, _vp(w), _vp(o)), "gmx_lstm_get_weights")
        return w, o

    def forward(self, ppm, last_byte, stream=0):
        """Lstm::Predict(last_byte) at a byte boundary: (probs[256], lstm_prediction_context)."""
        x = np.ascontiguousarray(ppm, np.float32)
        assert x.shape == (256,)
        probs = np.zeros(256, np.float32)
        ctx = C.c_uint32(0)
        check(self.L.gmx_lstm_forward(self.h, stream, int(last_byte), _vp(x), _vp(probs), C.byref(ctx)),
              "gmx_lstm_forward")
        return probs, ctx.value

    def perceive(self, byte, stream=0):
        check(self.L.gmx_lstm_perceive(self.h, stream, int(byte)), "gmx_lstm_perceive")

    def export(self, stream=0):
        """(long, short): the LSTM section of the .long file and the model's stretch of .short."""
        nl, ns = C.c_size_t(0), C.c_size_t(0)
        check(self.L.gmx_lstm_export(self.h, stream, None, C.byref(nl), None, C.byref(ns)), "gmx_lstm_export")
        bl, bs = np.zeros(nl.value, np.uint8), np.zeros(ns.value, np.uint8)
        check(self.L.gmx_lstm_export(self.h, stream, _vp(bl), C.byref(nl), _vp(bs), C.byref(ns)), "gmx_lstm_export")
        return bl[:nl.value].tobytes(), bs[:ns.value].tobytes()

    def import_(self, long_bytes, short_bytes, stream=0):
        a = np.frombuffer(long_bytes, np.uint8).copy()
        b = np.frombuffer(short_bytes, np.uint8).copy()
        check(self.L.gmx_lstm_import(self.h, stream, _vp(a), len(a), _vp(b), len(b)), "gmx_lstm_import")

    def copy_from(self, src, src_stream=0, stream=0):
        check(self.L.gmx_lstm_copy(self.h, stream, src.h, src_stream), "gmx_lstm_copy")

    def memory_usage(self):
        v = C.c_uint64(0)
        check(self.L.gmx_lstm_memory_usage(self.h, C.byref(v)), "gmx_lstm_memory_usage")
        return v.value

    def feed(self, batch, n_bytes, mixer_batch=None, slot=1, mixer_ctx_col=-1, ind_batch=None, ind_ctx_col=0):
        check(self.L.gmx_lstm_feed(self.h, batch.h, n_bytes, mixer_batch.h if mixer_batch else None, slot,
                                   mixer_ctx_col, ind_batch.h if ind_batch else None, ind_ctx_col), "gmx_lstm_feed")

    def run(self, batch, n_bytes=None, learn=True, timed=False):
        n = batch.max_bytes if n_bytes is None else n_bytes
        ms = C.c_float(0)
        check(self.L.gmx_lstm_run(self.h, batch.h, n, 1 if learn else 0, C.byref(ms) if timed else None),
              "gmx_lstm_run")
        return ms.value if timed else None


    def run_ragged(self, batch, n_bytes, learn=True):
        n = np.ascontiguousarray(n_bytes, np.uint64)
        assert n.shape == (self.S,)
        check(self.L.gmx_lstm_run_ragged(self.h, batch.h, n.ctypes.data_as(C.POINTER(C.c_uint64)), 1 if learn else 0),
              "gmx_lstm_run_ragged")


class LstmBatch:
    def __init__(self, group, max_bytes):
        self.g = group
        self.L = group.L
        self.max_bytes = int(max_bytes)
        h = C.c_void_p()
        check(self.L.gmx_lstm_batch_create(C.byref(h), group.h, self.max_bytes), "gmx_lstm_batch_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.gmx_lstm_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, fn, dtype, shape):
        ptr = fn(self.h)
        if not ptr:
            raise GmxError(-2, fn.__name__)
        n = int(np.prod(shape))
        buf = (C.c_byte * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    @property
    def ppm(self):
        return self._view(self.L.gmx_lstm_batch_ppm, np.float32, (self.g.S, self.max_bytes, 256))

    @property
    def bytes(self):
        return self._view(self.L.gmx_lstm_batch_bytes, np.uint8, (self.g.S, self.max_bytes))

    @property
    def predictions(self):
        return self._view(self.L.gmx_lstm_batch_predictions, np.float32, (self.g.S, self.max_bytes, 8))

    @property
    def active(self):
        return self._view(self.L.gmx_lstm_batch_active, np.uint8, (self.g.S, self.max_bytes, 8))

    @property
    def contexts(self):
        return self._view(self.L.gmx_lstm_batch_contexts, np.uint32, (self.g.S, self.max_bytes))

    def upload(self, n_bytes=None):
        check(self.L.gmx_lstm_batch_upload(self.h, self.max_bytes if n_bytes is None else n_bytes), "gmx_lstm_batch_upload")

    def download(self, n_bytes=None):
        check(self.L.gmx_lstm_batch_download(self.h, self.max_bytes if n_bytes is None else n_bytes),
              "gmx_lstm_batch_download")

    def wait(self):
        check(self.L.gmx_lstm_batch_wait(self.h), "gmx_lstm_batch_wait")
