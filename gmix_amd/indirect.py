"""Host-side handles of the Indirect-model banks (gmx_indirect / gmx_ind_batch of
include/gmxmix.h): test and bench harness, like bank.py for the mixers."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GmxError, IndirectDesc, check


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class IndirectGroup:
    """S banks of K Indirect models.  models = [(table_size, learning_rate)] in construction order;
    slots = [(slot_indirect, slot_run_map)] (default 2i, 2i+1); ns_next / rm_next: the two 256x2
    next-state tables of ShortTermMemory's state machines."""

    def __init__(self, models, ns_next, rm_next, n_streams=1, device=0, slots=None):
        self.L = _lib.lib()
        self.models = [(int(t), float(np.float32(lr))) for t, lr in models]
        self.K = len(self.models)
        self.S = int(n_streams)
        self.slots = [(2 * i, 2 * i + 1) for i in range(self.K)] if slots is None else [tuple(s) for s in slots]
        descs = (IndirectDesc * self.K)(*[IndirectDesc(t, lr, a, b) for (t, lr), (a, b) in zip(self.models, self.slots)])
        ns = np.ascontiguousarray(ns_next, np.uint8).reshape(512)
        rm = np.ascontiguousarray(rm_next, np.uint8).reshape(512)
        h = C.c_void_p()
        check(self.L.gmx_indirect_create(C.byref(h), descs, self.K, _vp(ns), _vp(rm), self.S, device),
              "gmx_indirect_create")
        self.h = h

    def set_cu_mask(self, words=None):
        """Compute units this bank's kernels may use: 32-bit words, bit i = CU i (None: all)."""
        w = list(words) if words else []
        arr = (C.c_uint32 * max(1, len(w)))(*w)
        check(self.L.gmx_indirect_set_cu_mask(self.h, arr, len(w)), "gmx_indirect_set_cu_mask")

    def close(self):
        if getattr(self, "h", None):
            self.L.gmx_indirect_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def bank_bytes(self):
        return self.L.gmx_indirect_bank_bytes(self.h)

    def reset(self):
        check(self.L.gmx_indirect_reset(self.h), "gmx_indirect_reset")

    def sync(self):
        check(self.L.gmx_indirect_sync(self.h), "gmx_indirect_sync")

    def forward(self, contexts, bit_context, stream=0):
        c = np.ascontiguousarray(contexts, np.uint32)
        assert c.shape == (self.K,)
        pred = np.zeros(2 * self.K, np.float32)
        act = np.zeros(2 * self.K, np.uint8)
        check(self.L.gmx_indirect_forward(self.h, stream, _vp(c), int(bit_context), _vp(pred), _vp(act)),
              "gmx_indirect_forward")
        return pred, act

    def learn(self, bit, stream=0):
        check(self.L.gmx_indirect_learn(self.h, stream, int(bit)), "gmx_indirect_learn")

    def chain_forward(self, group, contexts, bit_context, predictions, active, mixer_contexts, stream=0):
        """gmx_chain_forward: this bank's Predict and the mixers' (`group`) as one call; `predictions` / `active`
        (indices) are the blackboard without the Indirect models.  Returns (p, mixer outputs, predictions[2K],
        active[2K])."""
        c = np.ascontiguousarray(contexts, np.uint32)
        pr = np.ascontiguousarray(predictions, np.float32)
        ac = np.ascontiguousarray(active, np.int32)
        mc = np.ascontiguousarray(mixer_contexts, np.uint32)
        assert c.shape == (self.K,) and pr.shape == (group.topo.n_inputs,) and mc.shape == (group.topo.n_mixers,)
        p = C.c_float()
        out = np.zeros(group.topo.n_mixers, np.float32)
        pred = np.zeros(2 * self.K, np.float32)
        act = np.zeros(2 * self.K, np.uint8)
        check(self.L.gmx_chain_forward(self.h, group.h, stream, _vp(c), int(bit_context), _vp(pr), _vp(ac), len(ac),
                                       _vp(mc), C.byref(p), _vp(out), _vp(pred), _vp(act)), "gmx_chain_forward")
        return p.value, out, pred, act

    def run(self, batch, n_bits=None, learn=True, into=None, timed=False):
        n_bits = batch.max_bits if n_bits is None else n_bits
        ms = C.c_float(0)
        check(self.L.gmx_indirect_run(self.h, batch.h, n_bits, 1 if learn else 0, into.h if into else None,
                                      C.byref(ms) if timed else None), "gmx_indirect_run")
        return ms.value if timed else None

    def run_ragged(self, batch, n_bits, learn=True, into=None):
        n = np.ascontiguousarray(n_bits, np.uint64)
        assert n.shape == (self.S,)
        check(self.L.gmx_indirect_run_ragged(self.h, batch.h, n.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             1 if learn else 0, into.h if into else None), "gmx_indirect_run_ragged")

    def export(self, stream=0):
        n = C.c_size_t(0)
        check(self.L.gmx_indirect_export(self.h, stream, None, C.byref(n)), "gmx_indirect_export")
        buf = np.zeros(max(1, n.value), np.uint8)
        check(self.L.gmx_indirect_export(self.h, stream, _vp(buf), C.byref(n)), "gmx_indirect_export")
        return buf[:n.value].tobytes()

    def import_(self, data, stream=0):
        b = np.frombuffer(data, np.uint8)
        check(self.L.gmx_indirect_import(self.h, stream, _vp(b), len(b)), "gmx_indirect_import")

    def copy_from(self, src, src_stream=0, dst_stream=0):
        check(self.L.gmx_indirect_copy(self.h, dst_stream, src.h, src_stream), "gmx_indirect_copy")

    def slot_values(self, stream=0):
        """What the models' two blackboard slots hold ([2i] indirect, [2i+1] run map)."""
        v = np.zeros(2 * self.K, np.float32)
        check(self.L.gmx_indirect_slots_get(self.h, stream, v.ctypes.data_as(C.POINTER(C.c_float))), "gmx_indirect_slots_get")
        return v

    def set_slot_values(self, values, stream=0):
        v = np.ascontiguousarray(values, np.float32)
        assert v.shape == (2 * self.K,)
        check(self.L.gmx_indirect_slots_set(self.h, stream, v.ctypes.data_as(C.POINTER(C.c_float))), "gmx_indirect_slots_set")

    def memory_usage(self, model):
        v = C.c_uint64(0)
        check(self.L.gmx_indirect_memory_usage(self.h, model, C.byref(v)), "gmx_indirect_memory_usage")
        return v.value


class IndirectBatch:
    def __init__(self, group, max_bits):
        self.g = group
        self.L = group.L
        self.max_bits = int(max_bits)
        h = C.c_void_p()
        check(self.L.gmx_ind_batch_create(C.byref(h), group.h, self.max_bits), "gmx_ind_batch_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.gmx_ind_batch_destroy(self.h)
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
    def contexts(self):
        return self._view(self.L.gmx_ind_batch_contexts, np.uint32, (self.g.S, self.max_bits, self.g.K))

    @property
    def bit_contexts(self):
        return self._view(self.L.gmx_ind_batch_bit_contexts, np.uint32, (self.g.S, self.max_bits))

    @property
    def bits(self):
        return self._view(self.L.gmx_ind_batch_bits, np.uint8, (self.g.S, self.max_bits))

    @property
    def predictions(self):
        return self._view(self.L.gmx_ind_batch_predictions, np.float32, (self.g.S, self.max_bits, 2 * self.g.K))

    @property
    def active(self):
        return self._view(self.L.gmx_ind_batch_active, np.uint8, (self.g.S, self.max_bits, 2 * self.g.K))

    def set_records(self, stream, contexts, bit_contexts, bits):
        T = len(bits)
        self.contexts[stream, :T] = contexts
        self.bit_contexts[stream, :T] = bit_contexts
        self.bits[stream, :T] = bits

    def upload(self, n_bits=None):
        check(self.L.gmx_ind_batch_upload(self.h, self.max_bits if n_bits is None else n_bits), "gmx_ind_batch_upload")

    def download(self, n_bits=None):
        check(self.L.gmx_ind_batch_download(self.h, self.max_bits if n_bits is None else n_bits),
              "gmx_ind_batch_download")

    def wait(self):
        check(self.L.gmx_ind_batch_wait(self.h), "gmx_ind_batch_wait")

    def fill_synthetic(self, n_bits=None, seed=0, restart=True, ctx_mod=(0, 0, 0, 0)):
        m = np.asarray(ctx_mod, np.uint32)
        check(self.L.gmx_ind_batch_fill_synthetic(self.h, self.max_bits if n_bits is None else n_bits, seed,
                                                  1 if restart else 0, _vp(m)), "gmx_ind_batch_fill_synthetic")
