"""gmix_amd -- gmix's mixer hot path on AMD MI355X (gfx950): hand-written HIP kernels behind
a C ABI (include/gmxmix.h, libgmxmix.so), plus thin host-side mirrors of the reference's
Mixer / Predictor surface for that path.  Nothing here computes on the CPU."""
from . import topology
from ._lib import ABI_SYMBOLS, LIB_PATH, GmxError, build
from .bank import Batch, ChainStep, Lockstep, MixerGroup, Topology, device_count
from .indirect import IndirectBatch, IndirectGroup
from .lstm import LstmBatch, LstmGroup

__all__ = ["topology", "ABI_SYMBOLS", "LIB_PATH", "GmxError", "build", "Batch", "MixerGroup",
           "Lockstep", "ChainStep", "Topology", "device_count", "IndirectGroup", "IndirectBatch", "LstmGroup", "LstmBatch"]
