"""Python-side mirror of the mixer surface over the C ABI (numpy in, numpy out).

`Topology` is what `Predictor`'s constructor fixes before/while `AddMixers` runs
(predictor.cpp:17-40, 251-358); `MixerGroup` is S banks of it on one MI355X; `Batch` holds
records for the batched path.  All compute happens in libgmxmix.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GmxError, MixerDesc, TopologyStruct, check

BATCH_OUTPUTS = 1
BATCH_MASK = 2


class Topology:
    def __init__(self, n_inputs, mixers, skip=(1,)):
        """mixers: [(layer, table_size, learning_rate), ...] in construction order."""
        self.n_inputs = int(n_inputs)
        self.mixers = [(int(l), int(t), float(np.float32(lr))) for l, t, lr in mixers]
        self.skip = [int(i) for i in skip]
        self.l0 = sum(1 for m in self.mixers if m[0] == 0)
        self.l1 = sum(1 for m in self.mixers if m[0] == 1)
        self.has_final = any(m[0] == 2 for m in self.mixers)

    @property
    def n_mixers(self):
        return len(self.mixers)

    def weight_sizes(self):
        """weight_size_ of every mixer (mixer.cpp:17-26)."""
        out, k0, k1 = [], 0, 0
        for layer, _, _ in self.mixers:
            if layer == 0:
                out.append(self.n_inputs + k0)
                k0 += 1
            elif layer == 1:
                out.append(self.l0 + k1 + len(self.skip))
                k1 += 1
            else:
                out.append(self.l0 + self.l1 + len(self.skip))
        return out

    def bytes_per_bit(self):
        """Algorithmic HBM bytes per coded bit:
        8*W (each weight read once, written once) + 4*N inputs + 4*M contexts + 4 (p out)."""
        return 8 * sum(self.weight_sizes()) + 4 * self.n_inputs + 4 * self.n_mixers + 4

    def _struct(self):
        self._descs = (MixerDesc * len(self.mixers))(*[MixerDesc(l, t, lr) for l, t, lr in self.mixers])
        self._skip = (C.c_int32 * max(1, len(self.skip)))(*self.skip)
        return TopologyStruct(self.n_inputs, len(self.skip), self._skip, len(self.mixers), self._descs)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class MixerGroup:
    """S independent mixer banks on one device (gmx_group)."""

    def __init__(self, topo, n_streams=1, device=0):
        self.topo = topo
        self.S = n_streams
        self.L = _lib.lib()
        h = C.c_void_p()
        st = topo._struct()
        check(self.L.gmx_group_create(C.byref(h), C.byref(st), n_streams, device), "gmx_group_create")
        self.h = h

    def set_cu_mask(self, words=None):
        """Compute units this bank's kernels may use: 32-bit words, bit i = CU i (None: all)."""
        w = list(words) if words else []
        arr = (C.c_uint32 * max(1, len(w)))(*w)
        check(self.L.gmx_group_set_cu_mask(self.h, arr, len(w)), "gmx_group_set_cu_mask")

    def close(self):
        if getattr(self, "h", None):
            self.L.gmx_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self.L.gmx_group_reset(self.h), "gmx_group_reset")

    def sync(self):
        check(self.L.gmx_group_sync(self.h), "gmx_group_sync")

    @property
    def bank_bytes(self):
        return self.L.gmx_group_bank_bytes(self.h)

    def timer_start(self):
        check(self.L.gmx_group_timer_start(self.h), "gmx_group_timer_start")

    def timer_stop(self):
        ms = C.c_float(0)
        check(self.L.gmx_group_timer_stop(self.h, C.byref(ms)), "gmx_group_timer_stop")
        return ms.value

    # ---- per-bit surface (Predict / Perceive+Learn) ----
    def forward(self, predictions, active, contexts, stream=0, want_all=True):
        pred = np.ascontiguousarray(predictions, np.float32)
        ctx = np.ascontiguousarray(contexts, np.uint32)
        assert pred.shape == (self.topo.n_inputs,) and ctx.shape == (self.topo.n_mixers,)
        p = C.c_float()
        out = np.zeros(self.topo.n_mixers, np.float32) if want_all else None
        if active is None:
            act, na = None, -1
        else:
            act = np.ascontiguousarray(active, np.int32)
            na = len(act)
        check(self.L.gmx_bank_forward(self.h, stream, _vp(pred), _vp(act), na, _vp(ctx), C.byref(p),
                                      _vp(out)), "gmx_bank_forward")
        return p.value, out

    def learn(self, bit, stream=0):
        check(self.L.gmx_bank_learn(self.h, stream, int(bit)), "gmx_bank_learn")

    # ---- batched surface ----
    def run(self, batch, n_bits=None, learn=True, timed=False):
        n_bits = batch.max_bits if n_bits is None else n_bits
        ms = C.c_float(0)
        check(self.L.gmx_group_run(self.h, batch.h, n_bits, 1 if learn else 0,
                                   C.byref(ms) if timed else None), "gmx_group_run")
        return ms.value if timed else None

    def run_ragged(self, batch, n_bits, learn=True):
        """Stream s runs bits [0, n_bits[s]) of its records (gmx_group_run_ragged)."""
        n = np.ascontiguousarray(n_bits, np.uint64)
        assert n.shape == (self.S,)
        check(self.L.gmx_group_run_ragged(self.h, batch.h, n.ctypes.data_as(C.POINTER(C.c_uint64)), 1 if learn else 0),
              "gmx_group_run_ragged")

    # ---- persistence ----
    def export(self, stream=0):
        """(long_bytes, short_bytes) in the reference's checkpoint format."""
        nl, ns = C.c_size_t(0), C.c_size_t(0)
        check(self.L.gmx_bank_export(self.h, stream, None, C.byref(nl), None, C.byref(ns)),
              "gmx_bank_export(size)")
        lb = np.zeros(max(nl.value, 1), np.uint8)
        sb = np.zeros(max(ns.value, 1), np.uint8)
        check(self.L.gmx_bank_export(self.h, stream, _vp(lb), C.byref(nl), _vp(sb), C.byref(ns)),
              "gmx_bank_export")
        return lb.tobytes()[:nl.value], sb.tobytes()[:ns.value]

    def import_(self, long_bytes, short_bytes, stream=0):
        lb = np.frombuffer(long_bytes, np.uint8) if len(long_bytes) else np.zeros(1, np.uint8)
        sb = np.frombuffer(short_bytes, np.uint8)
        check(self.L.gmx_bank_import(self.h, stream, _vp(lb), len(long_bytes), _vp(sb), len(short_bytes)),
              "gmx_bank_import")

    def copy_from(self, src, src_stream=0, dst_stream=0):
        check(self.L.gmx_bank_copy(self.h, dst_stream, src.h, src_stream), "gmx_bank_copy")

    def memory_usage(self, mixer, stream=0):
        v = C.c_uint64()
        check(self.L.gmx_bank_memory_usage(self.h, stream, mixer, C.byref(v)), "gmx_bank_memory_usage")
        return v.value


class Batch:
    """Records for up to max_bits bits of every stream of a group (gmx_batch)."""

    def __init__(self, group, max_bits, outputs=True, mask=True, last_outputs=False):
        self.g = group
        self.L = group.L
        self.max_bits = int(max_bits)
        self.flags = (BATCH_OUTPUTS if outputs else 0) | (BATCH_MASK if mask else 0) | (8 if last_outputs else 0)
        h = C.c_void_p()
        check(self.L.gmx_batch_create(C.byref(h), group.h, self.max_bits, self.flags), "gmx_batch_create")
        self.h = h
        self.n_pad = self.L.gmx_batch_n_pad(h)
        self.mask_words = self.L.gmx_batch_mask_words(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.gmx_batch_destroy(self.h)
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
    def predictions(self):
        return self._view(self.L.gmx_batch_predictions, np.float32, (self.g.S, self.max_bits, self.n_pad))

    @property
    def active_mask(self):
        return self._view(self.L.gmx_batch_active_mask, np.uint32, (self.g.S, self.max_bits, self.mask_words))

    @property
    def contexts(self):
        return self._view(self.L.gmx_batch_contexts, np.uint32, (self.g.S, self.max_bits, self.g.topo.n_mixers))

    @property
    def bits(self):
        return self._view(self.L.gmx_batch_bits, np.uint8, (self.g.S, self.max_bits))

    @property
    def p(self):
        return self._view(self.L.gmx_batch_p, np.float32, (self.g.S, self.max_bits))

    @property
    def last_outputs(self):
        """[S][M]: every mixer's output of each stream's last bit of the run (GMX_BATCH_LAST_OUTPUTS)."""
        return self._view(self.L.gmx_batch_last_outputs, np.float32, (self.g.S, self.g.topo.n_mixers))

    @property
    def outputs(self):
        return self._view(self.L.gmx_batch_outputs, np.float32, (self.g.S, self.max_bits, self.g.topo.n_mixers))

    def set_records(self, stream, predictions, active, contexts, bits):
        """Fill stream `stream` from raw blackboard arrays: predictions[T,N] (stale slots allowed),
        active[T,N] flags (or None = all active), contexts[T,M], bits[T]."""
        T = len(bits)
        N = self.g.topo.n_inputs
        self.predictions[stream, :T, :N] = predictions
        self.predictions[stream, :T, N:] = 0
        if self.flags & BATCH_MASK:
            a = np.ones((T, N), np.uint8) if active is None else np.asarray(active, np.uint8)
            padded = np.zeros((T, self.mask_words * 32), np.uint8)
            padded[:, :N] = a != 0
            words = np.packbits(padded.reshape(T, self.mask_words, 32), axis=2, bitorder="little")
            self.active_mask[stream, :T, :] = words.view(np.uint32).reshape(T, self.mask_words)
        elif active is not None:
            # without a mask the caller must zero silent slots itself (exact: they add nothing)
            self.predictions[stream, :T, :N] = np.where(np.asarray(active) != 0, predictions, 0)
        self.contexts[stream, :T, :] = contexts
        self.bits[stream, :T] = bits

    def upload(self, n_bits=None):
        check(self.L.gmx_batch_upload(self.h, self.max_bits if n_bits is None else n_bits), "gmx_batch_upload")

    def download(self, n_bits=None):
        check(self.L.gmx_batch_download(self.h, self.max_bits if n_bits is None else n_bits),
              "gmx_batch_download")

    def wait(self):
        check(self.L.gmx_batch_wait(self.h), "gmx_batch_wait")

    def fill_synthetic(self, n_bits=None, seed=0, restart=True, ctx_mode=0, ctx_mod=1, zero_mod=0,
                       bit_mode=0):
        check(self.L.gmx_batch_fill_synthetic(self.h, self.max_bits if n_bits is None else n_bits, seed,
                                              1 if restart else 0, ctx_mode, ctx_mod, zero_mod, bit_mode),
              "gmx_batch_fill_synthetic")


def device_count():
    n = C.c_int(0)
    check(_lib.lib().gmx_device_count(C.byref(n)), "gmx_device_count")
    return n.value


class Lockstep:
    """All streams of a group one bit at a time (gmx_lockstep): each half step one hipGraph, or -- `persistent`,
    where it applies -- S persistent waves behind one doorbell."""

    def __init__(self, group, outputs=False, persistent=False):
        self.g = group
        self.L = group.L
        h = C.c_void_p()
        flags = (BATCH_OUTPUTS if outputs else 0) | (4 if persistent else 0)  # GMX_LOCKSTEP_PERSISTENT
        check(self.L.gmx_lockstep_create(C.byref(h), group.h, flags), "gmx_lockstep_create")
        self.h = h
        self.persistent = bool(self.L.gmx_lockstep_is_persistent(h))
        # the record batch is owned by the lock-step object: a Batch view that never destroys it
        b = Batch.__new__(Batch)
        b.g, b.L, b.max_bits = group, group.L, 1
        b.flags = BATCH_MASK | (BATCH_OUTPUTS if outputs else 0)
        b.h = C.c_void_p(self.L.gmx_lockstep_batch(h))
        b.n_pad = self.L.gmx_batch_n_pad(b.h)
        b.mask_words = self.L.gmx_batch_mask_words(b.h)
        b.close = lambda: None
        self.batch = b

    def predict(self):
        check(self.L.gmx_lockstep_predict(self.h), "gmx_lockstep_predict")
        return self.batch.p[:, 0]

    def learn(self):
        check(self.L.gmx_lockstep_learn(self.h), "gmx_lockstep_learn")

    def learn_predict(self):
        """Learn the bits of the step before and predict the next records, one graph."""
        check(self.L.gmx_lockstep_learn_predict(self.h), "gmx_lockstep_learn_predict")
        return self.batch.p[:, 0]

    def close(self):
        if getattr(self, "h", None):
            self.batch.h = None
            self.L.gmx_lockstep_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


STEP_LEARN, STEP_PREDICT = 1, 2


class ChainStep:
    """S decoders in lock step through every device-side model (gmx_chainstep): one hipGraph per coded bit.  `indirect`
    / `lstm` may be None.  Fill the host views for the streams that take part, set `what`, call step()."""

    def __init__(self, group, indirect=None, lstm=None, lstm_slot=-1, mixer_ctx_col=-1, ind_ctx_col=-1):
        self.g, self.L = group, group.L
        h = C.c_void_p()
        check(self.L.gmx_chainstep_create(C.byref(h), group.h, indirect.h if indirect else None,
                                          lstm.h if lstm else None, lstm_slot, mixer_ctx_col, ind_ctx_col),
              "gmx_chainstep_create")
        self.h = h
        S, M = group.S, group.topo.n_mixers
        n_pad = (group.topo.n_inputs + 3) // 4 * 4
        mw = (group.topo.n_inputs + 31) // 32
        self.mask_words = mw

        def view(name, dtype, shape):
            ptr = getattr(self.L, "gmx_chainstep_" + name)(h)
            if not ptr:
                return None
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_char * n).from_address(ptr), dtype=dtype).reshape(shape)

        self.predictions = view("predictions", np.float32, (S, n_pad))
        self.active_mask = view("active_mask", np.uint32, (S, mw))
        self.contexts = view("contexts", np.uint32, (S, M))
        K = indirect.K if indirect else 0
        self.ind_contexts = view("ind_contexts", np.uint32, (S, K)) if indirect else None
        self.bit_contexts = view("bit_contexts", np.uint32, (S,)) if indirect else None
        self.ppm = view("ppm", np.float32, (S, 256)) if lstm else None
        self.bits = view("bits", np.uint8, (S,))
        self.what = view("what", np.uint8, (S,))
        self.p = view("p", np.float32, (S,))
        self.outputs = view("outputs", np.float32, (S, M))

    def set_active(self, stream, active):
        """active[N] flags -> the stream's mask words."""
        padded = np.zeros(self.mask_words * 32, np.uint8)
        padded[:len(active)] = np.asarray(active) != 0
        self.active_mask[stream] = np.packbits(padded.reshape(self.mask_words, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)

    def commit(self, stream):
        """Optional: the stream's inputs for the coming step are complete (moved to the device now where the host can
        store there; step() does it for streams nobody committed)."""
        check(self.L.gmx_chainstep_commit(self.h, stream), "gmx_chainstep_commit")

    def step(self):
        check(self.L.gmx_chainstep_step(self.h), "gmx_chainstep_step")

    def launch(self):
        """step() in two halves: queued when this returns ..."""
        check(self.L.gmx_chainstep_launch(self.h), "gmx_chainstep_launch")

    def wait(self):
        """... p and outputs in place when this does."""
        check(self.L.gmx_chainstep_wait(self.h), "gmx_chainstep_wait")

    def close(self):
        if getattr(self, "h", None):
            self.L.gmx_chainstep_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
