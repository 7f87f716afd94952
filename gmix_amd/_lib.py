"""ctypes binding of libgmxmix.so (include/gmxmix.h).  The library is the product: if it is
missing or cannot be loaded this module raises -- there is no Python or CPU fallback."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# GMX_LIB overrides the library file (an instrumented build of the same sources, for profiling)
LIB_PATH = os.environ.get("GMX_LIB") or os.path.join(_HERE, "libgmxmix.so")
_LIB = None


class GmxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        L = _LIB
        msg = L.gmx_strerror(status).decode() if L else str(status)
        extra = L.gmx_last_error().decode() if (L and status == -3) else ""
        super().__init__(f"{where}: {msg} [{status}] {extra}".strip())


class MixerDesc(C.Structure):
    _fields_ = [("layer", C.c_int32), ("table_size", C.c_uint32), ("learning_rate", C.c_float)]


class IndirectDesc(C.Structure):
    _fields_ = [("table_size", C.c_uint32), ("learning_rate", C.c_float), ("slot_indirect", C.c_int32),
                ("slot_run_map", C.c_int32)]


class TopologyStruct(C.Structure):
    _fields_ = [("n_inputs", C.c_int32), ("n_skip", C.c_int32), ("skip_index", C.POINTER(C.c_int32)),
                ("n_mixers", C.c_int32), ("mixers", C.POINTER(MixerDesc))]


def build(force=False):
    """Compile libgmxmix.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", src, "clean"])
    subprocess.check_call(["make", "-s", "-C", src, "all"])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP extension is the only implementation; nothing falls back to CPU)")
    L = C.CDLL(LIB_PATH)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32
    L.gmx_strerror.restype = C.c_char_p
    L.gmx_strerror.argtypes = [i32]
    L.gmx_last_error.restype = C.c_char_p
    L.gmx_build_info.restype = C.c_char_p
    L.gmx_device_count.argtypes = [C.POINTER(C.c_int)]
    L.gmx_device_pci_bus_id.argtypes = [i32, C.c_char_p, C.c_size_t]
    L.gmx_group_create.argtypes = [C.POINTER(vp), C.POINTER(TopologyStruct), i32, i32]
    L.gmx_group_destroy.argtypes = [vp]
    L.gmx_group_destroy.restype = None
    for f in (L.gmx_group_n_streams, L.gmx_group_n_mixers, L.gmx_group_n_inputs, L.gmx_group_reset,
              L.gmx_group_sync):
        f.argtypes = [vp]
    L.gmx_group_timer_start.argtypes = [vp]
    L.gmx_group_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.gmx_group_bank_bytes.argtypes = [vp]
    L.gmx_group_bank_bytes.restype = u64
    L.gmx_bank_forward.argtypes = [vp, i32, vp, vp, i32, vp, C.POINTER(C.c_float), vp]
    L.gmx_bank_learn.argtypes = [vp, i32, i32]
    L.gmx_batch_create.argtypes = [C.POINTER(vp), vp, u64, C.c_uint]
    L.gmx_batch_destroy.argtypes = [vp]
    L.gmx_batch_destroy.restype = None
    L.gmx_batch_n_pad.argtypes = [vp]
    L.gmx_batch_mask_words.argtypes = [vp]
    L.gmx_batch_max_bits.argtypes = [vp]
    L.gmx_batch_max_bits.restype = u64
    for name in ("gmx_batch_predictions", "gmx_batch_active_mask", "gmx_batch_contexts",
                 "gmx_batch_bits", "gmx_batch_p", "gmx_batch_outputs", "gmx_batch_last_outputs"):
        f = getattr(L, name)
        f.argtypes = [vp]
        f.restype = vp
    L.gmx_batch_upload.argtypes = [vp, u64]
    L.gmx_batch_download.argtypes = [vp, u64]
    L.gmx_batch_wait.argtypes = [vp]
    L.gmx_batch_fill_synthetic.argtypes = [vp, u64, u64, u64, i32, u32, u32, i32]
    L.gmx_group_run.argtypes = [vp, vp, u64, i32, C.POINTER(C.c_float)]
    L.gmx_group_run_ragged.argtypes = [vp, vp, C.POINTER(u64), i32]
    L.gmx_bank_export.argtypes = [vp, i32, vp, C.POINTER(C.c_size_t), vp, C.POINTER(C.c_size_t)]
    L.gmx_bank_import.argtypes = [vp, i32, vp, C.c_size_t, vp, C.c_size_t]
    L.gmx_bank_copy.argtypes = [vp, i32, vp, i32]
    L.gmx_bank_memory_usage.argtypes = [vp, i32, i32, C.POINTER(u64)]
    L.gmx_indirect_create.argtypes = [C.POINTER(vp), C.POINTER(IndirectDesc), i32, vp, vp, i32, i32]
    L.gmx_indirect_destroy.argtypes = [vp]
    L.gmx_indirect_destroy.restype = None
    for f in (L.gmx_indirect_n_streams, L.gmx_indirect_n_models, L.gmx_indirect_reset, L.gmx_indirect_sync):
        f.argtypes = [vp]
    L.gmx_indirect_bank_bytes.argtypes = [vp]
    L.gmx_indirect_bank_bytes.restype = u64
    L.gmx_indirect_forward.argtypes = [vp, i32, vp, u32, vp, vp]
    L.gmx_chain_forward.argtypes = [vp, vp, i32, vp, u32, vp, vp, i32, vp, vp, vp, vp, vp]
    L.gmx_indirect_learn.argtypes = [vp, i32, i32]
    L.gmx_ind_batch_create.argtypes = [C.POINTER(vp), vp, u64]
    L.gmx_ind_batch_destroy.argtypes = [vp]
    L.gmx_ind_batch_destroy.restype = None
    L.gmx_ind_batch_max_bits.argtypes = [vp]
    L.gmx_ind_batch_max_bits.restype = u64
    for name in ("gmx_ind_batch_contexts", "gmx_ind_batch_bit_contexts", "gmx_ind_batch_bits",
                 "gmx_ind_batch_predictions", "gmx_ind_batch_active"):
        f = getattr(L, name)
        f.argtypes = [vp]
        f.restype = vp
    L.gmx_ind_batch_upload.argtypes = [vp, u64]
    L.gmx_ind_batch_download.argtypes = [vp, u64]
    L.gmx_ind_batch_wait.argtypes = [vp]
    L.gmx_ind_batch_fill_synthetic.argtypes = [vp, u64, u64, u64, vp]
    L.gmx_indirect_run.argtypes = [vp, vp, u64, i32, vp, C.POINTER(C.c_float)]
    L.gmx_indirect_run_ragged.argtypes = [vp, vp, C.POINTER(u64), i32, vp]
    L.gmx_indirect_export.argtypes = [vp, i32, vp, C.POINTER(C.c_size_t)]
    L.gmx_indirect_import.argtypes = [vp, i32, vp, C.c_size_t]
    L.gmx_indirect_copy.argtypes = [vp, i32, vp, i32]
    L.gmx_indirect_memory_usage.argtypes = [vp, i32, C.POINTER(u64)]
    L.gmx_indirect_slots_get.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.gmx_indirect_slots_set.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.gmx_lstm_create.argtypes = [C.POINTER(vp), i32, i32]
    L.gmx_lstm_destroy.argtypes = [vp]
    L.gmx_lstm_destroy.restype = None
    for f in (L.gmx_lstm_n_streams, L.gmx_lstm_reset, L.gmx_lstm_sync):
        f.argtypes = [vp]
    L.gmx_lstm_bank_bytes.argtypes = [vp]
    L.gmx_lstm_bank_bytes.restype = u64
    L.gmx_lstm_set_weights.argtypes = [vp, i32, vp]
    L.gmx_lstm_get_weights.argtypes = [vp, i32, vp, vp]
    L.gmx_lstm_batch_create.argtypes = [C.POINTER(vp), vp, u64]
    L.gmx_lstm_batch_destroy.argtypes = [vp]
    L.gmx_lstm_batch_destroy.restype = None
    for name in ("gmx_lstm_batch_ppm", "gmx_lstm_batch_bytes", "gmx_lstm_batch_predictions",
                 "gmx_lstm_batch_active", "gmx_lstm_batch_contexts"):
        f = getattr(L, name)
        f.argtypes = [vp]
        f.restype = vp
    L.gmx_lstm_batch_upload.argtypes = [vp, u64]
    L.gmx_lstm_batch_download.argtypes = [vp, u64]
    L.gmx_lstm_batch_wait.argtypes = [vp]
    L.gmx_lstm_run.argtypes = [vp, vp, u64, i32, C.POINTER(C.c_float)]
    L.gmx_lstm_run_ragged.argtypes = [vp, vp, C.POINTER(u64), i32]
    L.gmx_lstm_forward.argtypes = [vp, i32, i32, vp, vp, C.POINTER(u32)]
    L.gmx_lstm_perceive.argtypes = [vp, i32, i32]
    L.gmx_lstm_feed.argtypes = [vp, vp, u64, vp, i32, i32, vp, i32]
    L.gmx_lstm_export.argtypes = [vp, i32, vp, C.POINTER(C.c_size_t), vp, C.POINTER(C.c_size_t)]
    L.gmx_lstm_import.argtypes = [vp, i32, vp, C.c_size_t, vp, C.c_size_t]
    L.gmx_lstm_copy.argtypes = [vp, i32, vp, i32]
    L.gmx_lstm_memory_usage.argtypes = [vp, C.POINTER(u64)]
    L.gmx_lockstep_create.argtypes = [C.POINTER(vp), vp, C.c_uint]
    L.gmx_lockstep_destroy.argtypes = [vp]
    L.gmx_lockstep_destroy.restype = None
    L.gmx_lockstep_batch.argtypes = [vp]
    L.gmx_lockstep_batch.restype = vp
    L.gmx_lockstep_predict.argtypes = [vp]
    L.gmx_lockstep_is_persistent.argtypes = [vp]
    L.gmx_lockstep_learn.argtypes = [vp]
    L.gmx_lockstep_learn_predict.argtypes = [vp]
    L.gmx_chainstep_create.argtypes = [C.POINTER(vp), vp, vp, vp, i32, i32, i32]
    L.gmx_chainstep_destroy.argtypes = [vp]
    L.gmx_chainstep_destroy.restype = None
    L.gmx_chainstep_n_streams.argtypes = [vp]
    L.gmx_chainstep_step.argtypes = [vp]
    L.gmx_chainstep_commit.argtypes = [vp, i32]
    L.gmx_chainstep_launch.argtypes = [vp]
    L.gmx_chainstep_wait.argtypes = [vp]
    for name in ("predictions", "active_mask", "contexts", "ind_contexts", "bit_contexts", "ppm", "bits", "what", "p",
                 "outputs"):
        f = getattr(L, "gmx_chainstep_" + name)
        f.argtypes = [vp]
        f.restype = vp
    for name in ("gmx_group_set_cu_mask", "gmx_indirect_set_cu_mask", "gmx_lstm_set_cu_mask"):
        getattr(L, name).argtypes = [vp, C.POINTER(u32), i32]
    L.gmx_debug_math_probe.argtypes = [i32, vp, vp, u64, i32]
    L.gmx_debug_math_range.argtypes = [i32, u64, u64, i32, C.POINTER(C.c_ulonglong)]
    _LIB = L
    return L


def check(status, where):
    if status != 0:
        raise GmxError(status, where)


# Every symbol include/gmxmix.h declares; tests assert the built library exports them all.
ABI_SYMBOLS = [
    "gmx_strerror", "gmx_last_error", "gmx_device_count", "gmx_device_pci_bus_id", "gmx_build_info", "gmx_group_create",
    "gmx_group_destroy", "gmx_group_n_streams", "gmx_group_n_mixers", "gmx_group_n_inputs",
    "gmx_group_bank_bytes", "gmx_group_reset", "gmx_group_sync", "gmx_group_timer_start", "gmx_group_timer_stop", "gmx_bank_forward", "gmx_bank_learn",
    "gmx_batch_create", "gmx_batch_destroy", "gmx_batch_n_pad", "gmx_batch_mask_words",
    "gmx_batch_max_bits", "gmx_batch_predictions", "gmx_batch_active_mask", "gmx_batch_contexts",
    "gmx_batch_bits", "gmx_batch_p", "gmx_batch_outputs", "gmx_batch_last_outputs", "gmx_batch_upload", "gmx_batch_download",
    "gmx_batch_wait", "gmx_batch_fill_synthetic", "gmx_group_run", "gmx_group_run_ragged", "gmx_bank_export",
    "gmx_bank_import", "gmx_bank_copy", "gmx_bank_memory_usage",
    "gmx_lockstep_create", "gmx_lockstep_destroy", "gmx_lockstep_batch", "gmx_lockstep_is_persistent", "gmx_lockstep_predict", "gmx_lockstep_learn", "gmx_lockstep_learn_predict",
    "gmx_indirect_create", "gmx_indirect_destroy", "gmx_indirect_n_streams", "gmx_indirect_n_models",
    "gmx_indirect_bank_bytes", "gmx_indirect_reset", "gmx_indirect_sync", "gmx_indirect_forward", "gmx_chain_forward",
    "gmx_indirect_learn", "gmx_ind_batch_create", "gmx_ind_batch_destroy", "gmx_ind_batch_max_bits",
    "gmx_ind_batch_contexts", "gmx_ind_batch_bit_contexts", "gmx_ind_batch_bits",
    "gmx_ind_batch_predictions", "gmx_ind_batch_active", "gmx_ind_batch_upload", "gmx_ind_batch_download",
    "gmx_ind_batch_wait", "gmx_ind_batch_fill_synthetic", "gmx_indirect_run", "gmx_indirect_run_ragged", "gmx_indirect_export",
    "gmx_indirect_import", "gmx_indirect_copy", "gmx_indirect_memory_usage", "gmx_indirect_slots_get", "gmx_indirect_slots_set",
    "gmx_lstm_create", "gmx_lstm_destroy", "gmx_lstm_n_streams", "gmx_lstm_bank_bytes", "gmx_lstm_reset",
    "gmx_lstm_sync", "gmx_lstm_set_weights", "gmx_lstm_get_weights", "gmx_lstm_batch_create",
    "gmx_lstm_batch_destroy", "gmx_lstm_batch_ppm", "gmx_lstm_batch_bytes", "gmx_lstm_batch_predictions",
    "gmx_lstm_batch_active", "gmx_lstm_batch_contexts", "gmx_lstm_batch_upload", "gmx_lstm_batch_download",
    "gmx_lstm_batch_wait", "gmx_lstm_run", "gmx_lstm_run_ragged", "gmx_lstm_forward", "gmx_lstm_perceive", "gmx_lstm_feed",
    "gmx_lstm_export", "gmx_lstm_import", "gmx_lstm_copy", "gmx_lstm_memory_usage",
    "gmx_chainstep_create", "gmx_chainstep_destroy", "gmx_chainstep_n_streams", "gmx_chainstep_predictions",
    "gmx_chainstep_active_mask", "gmx_chainstep_contexts", "gmx_chainstep_ind_contexts", "gmx_chainstep_bit_contexts",
    "gmx_chainstep_ppm", "gmx_chainstep_bits", "gmx_chainstep_what", "gmx_chainstep_p", "gmx_chainstep_outputs",
    "gmx_chainstep_commit", "gmx_chainstep_step", "gmx_chainstep_launch", "gmx_chainstep_wait",
    "gmx_group_set_cu_mask", "gmx_indirect_set_cu_mask", "gmx_lstm_set_cu_mask",
]
