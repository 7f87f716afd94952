"""ctypes front-end of the CPU oracle (oracle/gmx_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
It also reads the GMXD dump files written by oracle/ref_build/ref_mixer_harness (the real
reference), which is how the oracle itself is pinned (tests/test_oracle.py).
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C restatement (and, where /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgmxoracle.so")
        src = [os.path.join(_HERE, f) for f in ("gmx_oracle.c", "gmx_oracle_ind.c", "gmx_synth.h", "gmx_ind_synth.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
        L = C.CDLL(path)
        L.gmxo_create.restype = C.c_void_p
        L.gmxo_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p]
        L.gmxo_destroy.argtypes = [C.c_void_p]
        L.gmxo_predict.restype = C.c_float
        L.gmxo_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p]
        L.gmxo_learn.argtypes = [C.c_void_p, C.c_int]
        L.gmxo_run.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        for f in (L.gmxo_export_short, L.gmxo_export_long):
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gmxo_memory_usage.restype = C.c_uint64
        L.gmxo_memory_usage.argtypes = [C.c_void_p, C.c_int]
        L.gmxo_encode.restype = C.c_size_t
        L.gmxo_encode.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.gmxo_synth_fill.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                      C.c_uint32, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.gmxo_logistic.restype = C.c_float
        L.gmxo_logistic.argtypes = [C.c_float]
        L.gmxo_squash_clamp.restype = C.c_float
        L.gmxo_squash_clamp.argtypes = [C.c_float]
        L.gmxo_decay_base.restype = C.c_float
        L.gmxo_decay_base.argtypes = [C.c_uint64]
        L.gmxo_libm_expf_array.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gmxo_discretize.restype = C.c_uint32
        L.gmxo_discretize.argtypes = [C.c_float]
        L.gmxo_stream_new.restype = C.c_void_p
        L.gmxo_stream_new.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                      C.c_int]
        L.gmxo_stream_next.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.gmxo_stream_free.argtypes = [C.c_void_p]
        L.gmxo_fnv64.restype = C.c_uint64
        L.gmxo_fnv64.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def synth(n, m, T, seed=0, ctx_mode=0, ctx_mod=1, zero_mod=0, bit_mode=0):
    """Materialise the synthetic stream: (pred[T,n] raw, active[T,n] u8, ctx[T,m] u32, bits[T] u8)."""
    pred = np.zeros((T, n), np.float32)
    active = np.zeros((T, n), np.uint8)
    ctx = np.zeros((T, m), np.uint32)
    bits = np.zeros(T, np.uint8)
    lib().gmxo_synth_fill(seed, n, m, ctx_mode, ctx_mod, zero_mod, bit_mode, T, _p(pred), _p(active),
                          _p(ctx), _p(bits))
    return pred, active, ctx, bits


class Stream:
    """The same synthetic stream, generated chunk by chunk (long runs)."""

    def __init__(self, n, m, seed=0, ctx_mode=0, ctx_mod=1, zero_mod=0, bit_mode=0):
        self.n, self.m = n, m
        self.h = lib().gmxo_stream_new(seed, n, m, ctx_mode, ctx_mod, zero_mod, bit_mode)

    def next(self, T):
        pred = np.zeros((T, self.n), np.float32)
        active = np.zeros((T, self.n), np.uint8)
        ctx = np.zeros((T, self.m), np.uint32)
        bits = np.zeros(T, np.uint8)
        lib().gmxo_stream_next(self.h, T, _p(pred), _p(active), _p(ctx), _p(bits))
        return pred, active, ctx, bits

    def __del__(self):
        if getattr(self, "h", None):
            lib().gmxo_stream_free(self.h)
            self.h = None


class Bank:
    """One predictor's mixers, CPU restatement (dense tables)."""

    def __init__(self, n, skip, topo):
        """topo: list of (layer, table_size, lr) in construction order; skip: list of model indices."""
        self.n, self.m = n, len(topo)
        self.skip = np.asarray(list(skip), np.int32)
        layer = np.asarray([t[0] for t in topo], np.int32)
        table = np.asarray([t[1] for t in topo], np.uint32)
        lr = np.asarray([t[2] for t in topo], np.float32)
        self.h = lib().gmxo_create(n, len(self.skip), _p(self.skip), self.m, _p(layer), _p(table),
                                   _p(lr))

    def __del__(self):
        if getattr(self, "h", None):
            lib().gmxo_destroy(self.h)
            self.h = None

    def predict(self, predictions, active_idx, ctx):
        predictions = np.ascontiguousarray(predictions, np.float32)
        active_idx = np.ascontiguousarray(active_idx, np.int32)
        ctx = np.ascontiguousarray(ctx, np.uint32)
        out = np.zeros(self.m, np.float32)
        p = lib().gmxo_predict(self.h, _p(predictions), _p(active_idx), len(active_idx), _p(ctx),
                               _p(out))
        return p, out

    def learn(self, bit):
        lib().gmxo_learn(self.h, int(bit))

    def run(self, pred, active, ctx, bits, nolearn_from=None, want_all=True):
        T = len(bits)
        pred = np.ascontiguousarray(pred, np.float32)
        active = np.ascontiguousarray(active, np.uint8)
        ctx = np.ascontiguousarray(ctx, np.uint32)
        bits = np.ascontiguousarray(bits, np.uint8)
        p = np.zeros(T, np.float32)
        outs = np.zeros((T, self.m), np.float32) if want_all else None
        nl = (1 << 64) - 1 if nolearn_from is None else nolearn_from
        lib().gmxo_run(self.h, T, _p(pred), _p(active), _p(ctx), _p(bits), nl, _p(p), _p(outs))
        return p, outs

    def export_short(self):
        n = lib().gmxo_export_short(self.h, None, 0)
        buf = np.zeros(n, np.uint8)
        lib().gmxo_export_short(self.h, _p(buf), n)
        return buf.tobytes()

    def export_long(self):
        n = lib().gmxo_export_long(self.h, None, 0)
        buf = np.zeros(max(n, 1), np.uint8)
        lib().gmxo_export_long(self.h, _p(buf), n)
        return buf.tobytes()[:n]

    def memory_usage(self, j):
        return lib().gmxo_memory_usage(self.h, j)


def encode(bits, p):
    """Arithmetic-code bits with probabilities p (coder/encoder.cpp restated); returns bytes."""
    bits = np.ascontiguousarray(bits, np.uint8)
    p = np.ascontiguousarray(p, np.float32)
    n = lib().gmxo_encode(len(bits), _p(bits), _p(p), None, 0)
    out = np.zeros(n, np.uint8)
    lib().gmxo_encode(len(bits), _p(bits), _p(p), _p(out), n)
    return out.tobytes()


def fnv64(outs, p, h0=0):
    """The harness's strong checksum: FNV-1a-style over out_all[t,:] then p[t], per bit."""
    outs = np.ascontiguousarray(outs, np.float32)
    p = np.ascontiguousarray(p, np.float32)
    return lib().gmxo_fnv64(h0, len(p), outs.shape[1], _p(outs), _p(p))


def read_dump(path):
    """Parse a GMXD file from ref_mixer_harness into a dict."""
    with open(path, "rb") as f:
        b = f.read()
    magic, ver, n, m, l0, l1, has_final, n_skip = struct.unpack_from("<8I", b, 0)
    assert magic == 0x44584D47 and ver == 1
    T, dump = struct.unpack_from("<2Q", b, 32)
    off = 48
    rec = np.frombuffer(b, np.float32, dump * (m + 1), off).reshape(dump, m + 1)
    off += 4 * dump * (m + 1)
    h32, = struct.unpack_from("<I", b, off)
    off += 4
    acc, = struct.unpack_from("<d", b, off)
    off += 8
    h64, = struct.unpack_from("<Q", b, off)
    off += 8
    ns, = struct.unpack_from("<Q", b, off)
    off += 8
    short = b[off:off + ns]
    off += ns
    nl, = struct.unpack_from("<Q", b, off)
    off += 8
    long_ = b[off:off + nl]
    off += nl
    mem = np.frombuffer(b, np.uint64, m, off).copy()
    return dict(n=n, m=m, l0=l0, l1=l1, has_final=has_final, n_skip=n_skip, T=T, dump=dump,
                outs=rec[:, :m].copy(), p=rec[:, m].copy(), h32=h32, acc=acc, h64=h64,
                short=short, long=long_, mem=mem)


# ---- Indirect models (oracle/gmx_oracle_ind.c) -------------------
def _ind_lib():
    L = lib()
    if not getattr(L, "_ind_ready", False):
        L.gmxo_ind_create.restype = C.c_void_p
        L.gmxo_ind_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gmxo_ind_destroy.argtypes = [C.c_void_p]
        L.gmxo_ind_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.gmxo_ind_learn.argtypes = [C.c_void_p, C.c_int]
        L.gmxo_ind_run.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                   C.c_void_p, C.c_void_p]
        L.gmxo_ind_memory_usage.restype = C.c_uint64
        L.gmxo_ind_memory_usage.argtypes = [C.c_void_p, C.c_int]
        L.gmxo_ind_export.restype = C.c_size_t
        L.gmxo_ind_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gmxo_ind_synth_fill.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        L.gmxo_ind_fnv64.restype = C.c_uint64
        L.gmxo_ind_fnv64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L._ind_ready = True
    return L


def ind_synth(k, T, seed=0, ctx_mod=(0, 0, 0, 0)):
    """(ctx[T,k] u32, bit_context[T] u32, bits[T] u8) of oracle/gmx_ind_synth.h."""
    L = _ind_lib()
    ctx = np.zeros((T, k), np.uint32)
    bc = np.zeros(T, np.uint32)
    bits = np.zeros(T, np.uint8)
    mod = np.asarray(ctx_mod, np.uint32)
    L.gmxo_ind_synth_fill(seed, k, _p(mod), T, _p(ctx), _p(bc), _p(bits))
    return ctx, bc, bits


class IndirectBank:
    """K reference Indirect models: models = [(table_size, learning_rate), ...]."""

    def __init__(self, models, ns_next, rm_next):
        self.L = _ind_lib()
        self.k = len(models)
        ts = np.asarray([m[0] for m in models], np.uint32)
        lr = np.asarray([m[1] for m in models], np.float32)
        self._ns = np.ascontiguousarray(ns_next, np.uint8).reshape(512)
        self._rm = np.ascontiguousarray(rm_next, np.uint8).reshape(512)
        self.h = self.L.gmxo_ind_create(self.k, _p(ts), _p(lr), _p(self._ns), _p(self._rm))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.gmxo_ind_destroy(self.h)
            self.h = None

    def predict(self, ctx, bit_context):
        c = np.ascontiguousarray(ctx, np.uint32)
        pred = np.zeros(2 * self.k, np.float32)
        act = np.zeros(2 * self.k, np.uint8)
        self.L.gmxo_ind_predict(self.h, _p(c), int(bit_context), _p(pred), _p(act))
        return pred, act

    def learn(self, bit):
        self.L.gmxo_ind_learn(self.h, int(bit))

    def run(self, ctx, bit_context, bits, nolearn_from=None):
        T = len(bits)
        c = np.ascontiguousarray(ctx, np.uint32)
        bc = np.ascontiguousarray(bit_context, np.uint32)
        b = np.ascontiguousarray(bits, np.uint8)
        pred = np.zeros((T, 2 * self.k), np.float32)
        act = np.zeros((T, 2 * self.k), np.uint8)
        nl = (1 << 64) - 1 if nolearn_from is None else nolearn_from
        self.L.gmxo_ind_run(self.h, T, _p(c), _p(bc), _p(b), nl, _p(pred), _p(act))
        return pred, act

    def memory_usage(self, i):
        return int(self.L.gmxo_ind_memory_usage(self.h, i))

    def export(self):
        n = self.L.gmxo_ind_export(self.h, None, 0)
        buf = np.zeros(n, np.uint8)
        self.L.gmxo_ind_export(self.h, _p(buf), n)
        return buf.tobytes()


def ind_fnv64(pred, active, h0=0xcbf29ce484222325):
    """The harness' running checksum over (prediction bits, active flag) of every slot and bit."""
    L = _ind_lib()
    u = np.ascontiguousarray(pred, np.float32).reshape(-1)
    a = np.ascontiguousarray(active, np.uint8).reshape(-1)
    return int(L.gmxo_ind_fnv64(_p(u), _p(a), len(u), h0))


def read_ind_dump(path):
    """GMXI file of oracle/ref_build/ref_indirect_harness."""
    raw = open(path, "rb").read()
    magic, K, T, D = struct.unpack_from("<4I", raw, 0)
    assert magic == 0x49584D47
    o = 16
    ns_next = np.frombuffer(raw, np.uint8, 512, o).copy(); o += 512
    rm_next = np.frombuffer(raw, np.uint8, 512, o).copy(); o += 512
    pred = np.zeros((D, 2 * K), np.float32)
    act = np.zeros((D, 2 * K), np.uint8)
    for t in range(D):
        pred[t] = np.frombuffer(raw, np.float32, 2 * K, o); o += 8 * K
        act[t] = np.frombuffer(raw, np.uint8, 2 * K, o); o += 2 * K
    (h64,) = struct.unpack_from("<Q", raw, o); o += 8
    usage = np.frombuffer(raw, np.uint64, K, o).copy(); o += 8 * K
    (ll,) = struct.unpack_from("<Q", raw, o); o += 8
    long_b = raw[o:o + ll]
    assert o + ll == len(raw)
    return dict(K=K, T=T, D=D, ns_next=ns_next, rm_next=rm_next, pred=pred, active=act, h64=h64,
                usage=usage, long=long_b)


# ---- LSTM byte model (oracle/gmx_oracle_lstm.c) -------------------
def _lstm_lib():
    L = lib()
    if not getattr(L, "_lstm_ready", False):
        L.gmxo_lstm_create.restype = C.c_void_p
        L.gmxo_lstm_destroy.argtypes = [C.c_void_p]
        L.gmxo_lstm_get_weights.argtypes = [C.c_void_p, C.c_void_p]
        L.gmxo_lstm_set_weights.argtypes = [C.c_void_p, C.c_void_p]
        L.gmxo_lstm_model_predict.restype = C.c_int
        L.gmxo_lstm_model_predict.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p,
                                              C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p]
        L.gmxo_lstm_model_learn.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.gmxo_lstm_state_hash.restype = C.c_uint64
        L.gmxo_lstm_state_hash.argtypes = [C.c_void_p]
        L.gmxo_lstm_weights_hash.restype = C.c_uint64
        L.gmxo_lstm_weights_hash.argtypes = [C.c_void_p, C.c_int]
        L.gmxo_lstm_run_synth.restype = C.c_uint64
        L.gmxo_lstm_run_synth.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        L.gmxo_lstm_run_synth2.restype = C.c_uint64
        L.gmxo_lstm_run_synth2.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        L.gmxo_srand.argtypes = [C.c_uint]
        L.gmxo_lstm_synth_fill.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.gmxo_lstm_run.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_void_p]
        L.gmxo_lstm_get_output_layer.argtypes = [C.c_void_p, C.c_void_p]
        L.gmxo_lstm_predict_byte.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
        L.gmxo_lstm_perceive_byte.argtypes = [C.c_void_p, C.c_uint32]
        L.gmxo_lstm_update_steps.restype = C.c_uint64
        L.gmxo_lstm_update_steps.argtypes = [C.c_void_p]
        for f in (L.gmxo_lstm_export_short, L.gmxo_lstm_export_long):
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_void_p]
        for f in (L.gmxo_lstm_import_short, L.gmxo_lstm_import_long):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L._lstm_ready = True
    return L


class LstmModel:
    """The reference's LstmModel (Lstm(256, 256, 50, 1, 100, 0.03, 10)), weights initialised from
    rand() after srand(seed) exactly like the reference's constructor chain."""

    def __init__(self, srand_seed=0xDEADBEEF):
        self.L = _lstm_lib()
        self.L.gmxo_srand(srand_seed)
        self.h = self.L.gmxo_lstm_create()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.gmxo_lstm_destroy(self.h)
            self.h = None

    def weights_hash(self, with_output_layer=False):
        return int(self.L.gmxo_lstm_weights_hash(self.h, 1 if with_output_layer else 0))

    def weights(self):
        w = np.zeros((3, 50, 563), np.float32)
        self.L.gmxo_lstm_get_weights(self.h, _p(w))
        return w

    def output_layer(self):
        o = np.zeros((100, 256, 51), np.float32)
        self.L.gmxo_lstm_get_output_layer(self.h, _p(o))
        return o

    def export_short(self):
        """LstmModel::WriteToDisk: the model's stretch of the .short file."""
        buf = np.zeros(self.L.gmxo_lstm_export_short(self.h, None), np.uint8)
        self.L.gmxo_lstm_export_short(self.h, _p(buf))
        return buf.tobytes()

    def export_long(self):
        """The LSTM section of LongTermMemory::WriteToDisk (.long file)."""
        buf = np.zeros(self.L.gmxo_lstm_export_long(self.h, None), np.uint8)
        self.L.gmxo_lstm_export_long(self.h, _p(buf))
        return buf.tobytes()

    def import_state(self, long_bytes, short_bytes):
        a = np.frombuffer(long_bytes, np.uint8).copy()
        b = np.frombuffer(short_bytes, np.uint8).copy()
        assert self.L.gmxo_lstm_import_long(self.h, _p(a), len(a)) == 0
        assert self.L.gmxo_lstm_import_short(self.h, _p(b), len(b)) == 0

    def bits_from_probs(self, probs, byte):
        """The 8 bit predictions LstmModel::Predict derives from a byte distribution while `byte` is coded
        (lstm-model.cpp:34-48): (prediction[8], active[8], (top, mid, bot) after the eighth)."""
        L = self.L
        L.gmxo_lstm_bit_from_probs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        probs = np.ascontiguousarray(probs, np.float32)
        top, mid, bot = C.c_int(255), C.c_int(127), C.c_int(0)
        if not hasattr(self, "_lb"):
            self._lb, self._pr, self._cx = C.c_uint32(0), C.c_float(0), C.c_uint32(0)
        pr = C.c_float(self._pr.value)
        out, act = np.zeros(8, np.float32), np.zeros(8, np.uint8)
        for k in range(8):
            new_bit = (int(byte) >> (8 - k)) & 1 if k else 0
            act[k] = L.gmxo_lstm_bit_from_probs(_p(probs), int(k == 0), new_bit, C.byref(top), C.byref(mid), C.byref(bot),
                                                C.byref(pr))
            out[k] = pr.value
        self._pr.value = pr.value
        return out, act, (top.value, mid.value, bot.value)

    def predict_byte(self, ppm, last_byte):
        x = np.ascontiguousarray(ppm, np.float32)
        probs = np.zeros(256, np.float32)
        ctx = C.c_uint32(0)
        self.L.gmxo_lstm_predict_byte(self.h, _p(x), int(last_byte), _p(probs), C.byref(ctx))
        return probs, ctx.value

    def perceive_byte(self, byte):
        self.L.gmxo_lstm_perceive_byte(self.h, int(byte))

    def run(self, ppm, data, learn=True):
        """Whole bytes through LstmModel::Predict x 8 (+ Learn): (pred[n,8], active[n,8], ctx[n])."""
        ppm = np.ascontiguousarray(ppm, np.float32)
        data = np.ascontiguousarray(data, np.uint8)
        n = len(data)
        pred = np.zeros((n, 8), np.float32)
        act = np.zeros((n, 8), np.uint8)
        ctx = np.zeros(n, np.uint32)
        if not hasattr(self, "_lb"):
            self._lb, self._pr, self._cx = C.c_uint32(0), C.c_float(0), C.c_uint32(0)
        self.L.gmxo_lstm_run(self.h, n, _p(ppm), _p(data), 1 if learn else 0, C.byref(self._lb), C.byref(self._pr),
                             C.byref(self._cx), _p(pred), _p(act), _p(ctx))
        return pred, act, ctx

    def run_synth(self, n_bytes, seed=0, mask=255, dump=0, nolearn_from=None):
        """Drive the model with oracle/gmx_lstm_synth.h like the reference harness does; returns
        (fnv over all bits, predictions[dump,8], active[dump,8], context[dump])."""
        pred = np.zeros((dump, 8), np.float32)
        act = np.zeros((dump, 8), np.uint8)
        ctx = np.zeros(dump, np.uint32)
        h = self.L.gmxo_lstm_run_synth2(self.h, n_bytes, seed, mask, dump,
                                        (1 << 64) - 1 if nolearn_from is None else nolearn_from, _p(pred), _p(act), _p(ctx))
        return int(h), pred, act, ctx


def read_lstm_dump(path):
    """GMXL file of oracle/ref_build/ref_lstm_harness."""
    raw = open(path, "rb").read()
    magic, N, D = struct.unpack_from("<3I", raw, 0)
    assert magic == 0x4C584D47
    (hw,) = struct.unpack_from("<Q", raw, 12)
    o = 20
    pred = np.zeros((D, 8), np.float32)
    act = np.zeros((D, 8), np.uint8)
    ctx = np.zeros(D, np.uint32)
    for n in range(D):
        for k in range(8):
            pred[n, k], act[n, k] = struct.unpack_from("<fB", raw, o)
            o += 5
            if k == 0:
                (ctx[n],) = struct.unpack_from("<I", raw, o)
                o += 4
    h, hl, usage, short_size, hs = struct.unpack_from("<5Q", raw, o)
    o += 40
    top, mid, bot = struct.unpack_from("<3i", raw, o)
    probs = np.frombuffer(raw, np.float32, 256, o + 12).copy()
    assert o + 12 + 1024 == len(raw)
    return dict(N=N, D=D, init_weights_hash=hw, pred=pred, active=act, ctx=ctx, h64=h, long_hash=hl, usage=usage,
                short_size=short_size, short_hash=hs, top=top, mid=mid, bot=bot, probs=probs)


def lstm_synth(n_bytes, seed=0, mask=255):
    """(ppm[n,256] f32, bytes[n] u8) of oracle/gmx_lstm_synth.h."""
    L = _lstm_lib()
    ppm = np.zeros((n_bytes, 256), np.float32)
    data = np.zeros(n_bytes, np.uint8)
    L.gmxo_lstm_synth_fill(seed, mask, n_bytes, _p(ppm), _p(data))
    return ppm, data


def fnv64_bytes(a):
    L = _lstm_lib()
    L.gmxo_fnv64_bytes.restype = C.c_uint64
    L.gmxo_fnv64_bytes.argtypes = [C.c_void_p, C.c_uint64]
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(a, np.uint8)
    b = np.ascontiguousarray(a, np.uint8).reshape(-1)
    return int(L.gmxo_fnv64_bytes(_p(b), len(b)))
