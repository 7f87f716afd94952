"""The C-ABI library loads here (no GPU), exports every symbol include/gmxmix.h declares,
validates topologies on the host, and refuses to compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import gmix_amd
from gmix_amd import _lib, topology

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "gmxmix.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gmx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_exactly_the_header():
    L = C.CDLL(gmix_amd.LIB_PATH)
    declared = header_functions()
    assert declared == sorted(gmix_amd.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/gmxmix.h but not exported"


def test_strerror_and_build_info():
    L = _lib.lib()
    assert L.gmx_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.gmx_strerror(-4)
    assert b"gfx950" in L.gmx_build_info()


def test_topology_bookkeeping_matches_reference_counts():
    t = topology.stock(90)
    ws = t.weight_sizes()
    assert (t.l0, t.l1, t.has_final, t.n_mixers) == (24, 8, True, 33)
    assert ws[:24] == list(range(90, 114)) and ws[24:32] == list(range(25, 33)) and ws[32] == 33
    assert sum(ws) == 2697
    assert sum(m[1] for m in t.mixers) == 343676  # total gate rows
    assert t.bytes_per_bit() == 22072
    assert topology.single(256).bytes_per_bit() == 3080
    assert topology.synth3(256).bytes_per_bit() == 54608


@pytest.mark.skipif(gmix_amd.device_count() > 0, reason="CPU-only behaviour")
def test_no_device_means_no_compute():
    with pytest.raises(gmix_amd.GmxError) as e:
        gmix_amd.MixerGroup(topology.single(16, 8), 1)
    assert e.value.status == -4  # GMX_ERR_NO_DEVICE: the product never falls back to the CPU


def test_invalid_topologies_rejected():
    bad = [
        topology.Topology(16, [(1, 8, .1)], skip=()),                 # no layer-0 mixer
        topology.Topology(16, [(0, 8, .1), (2, 1, .1), (2, 1, .1)]),  # two finals
        topology.Topology(16, [(0, 8, .1), (1, 8, .1), (0, 8, .1)]),  # layers out of order
        topology.Topology(16, [(0, 0, .1)]),                          # empty table
        topology.Topology(16, [(0, 8, .1)] * 65),                     # too many mixers
        topology.Topology(4096, [(0, 8, .1)]),                        # too many inputs
        topology.Topology(16, [(0, 8, .1)], skip=(16,)),              # skip index out of range
        topology.Topology(16, [(0, 8, .1)] * 64 + []),                # fine count but ...
    ]
    for t in bad[:-1]:
        with pytest.raises(gmix_amd.GmxError) as e:
            gmix_amd.MixerGroup(t, 1)
        assert e.value.status == -1, t.mixers[:3]


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    """gmix_amd/host/gmx_mixer.h (the C++ mirror of Mixer / Predictor) builds with plain g++
    against the C ABI; running it needs a GPU (tests/test_gpu_host_cpp.py)."""
    import subprocess
    exe = str(tmp_path / "t")
    subprocess.check_call([
        "g++", "-std=c++17", "-Wall", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_host_adapter.cpp"),
        "-L" + os.path.join(ROOT, "gmix_amd"), "-lgmxmix", "-L" + os.path.join(ROOT, "oracle"), "-lgmxoracle",
        "-Wl,-rpath," + os.path.join(ROOT, "gmix_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
        "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)


def test_kernels_do_not_spill_to_scratch():
    """The bit-loop kernels issue their vector-memory instructions by hand and wait with counted
    s_waitcnt vmcnt(N): a compiler-inserted scratch spill (a VMEM instruction the count does
    not know about) would silently break that bookkeeping.  Assert the resource report."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    src = os.path.join(ROOT, "gmix_amd", "csrc")

    def report_of(f):
        out = subprocess.run(
            ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
             "-fno-gpu-flush-denormals-to-zero", "-c", os.path.join(src, f), "-o", "/dev/null",
             "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=src)
        return out.stderr + out.stdout

    # (gmx_indirect.hip: a table-entry register that the compiler parks in scratch is stored there the moment the
    # asm load has been ISSUED, not when its data is in -- it did, once, and the kernel read garbage)
    files = ("gmx_single.hip", "gmx_stock.hip", "gmx_kernels.hip", "gmx_wide.hip", "gmx_indirect.hip", "gmx_lstm.hip")
    with ThreadPoolExecutor(6) as ex:   # (six compilations side by side: 25 s instead of 80)
        reports = dict(zip(files, ex.map(report_of, files)))
    for f in files[:-1]:
        report = reports[f]
        scratch = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", report)
        vspill = re.findall(r"VGPRs Spill: (\d+)", report)
        assert scratch and all(s == "0" for s in scratch), (f, scratch)
        assert all(s == "0" for s in vspill), (f, vspill)
    # gmx_lstm.hip: the default batched build (two workgroups per CU) keeps its one in-flight load in an AGPR and
    # must not touch scratch; the three-per-CU tuning build (GMX_LSTM_BUILD=3) spills and is exempt, and so is the
    # per-byte session, which has no asm load in flight (its inputs come from the mailbox by ordinary loads)
    blocks = re.split(r"Function Name: ", reports["gmx_lstm.hip"])[1:]
    seen = 0
    for blk in blocks:
        if "gmx_lstm_kernelILi154ELi2ELb0E" in blk:
            seen += 1
            assert re.search(r"ScratchSize \[bytes/lane\]: 0\b", blk), blk[:200]
    assert seen == 1


def test_stock_kernels_leave_the_reserved_registers_alone():
    """gmx_stock.hip's instruction streams (gmx_stock_asm.inc) own v44..v255, a140..a255 and
    s70..s101 by convention: the kernels are compiled with register limits so that hipcc's own
    code stays below.  Check the ISA: no compiler-generated instruction in the reserved ranges,
    no scratch, and the committed .inc is what the generator produces."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_stock_regs
    bad, counts = check_stock_regs.check(check_stock_regs.compile_to_asm())
    assert not bad, bad[:5]
    assert any("session" in k for k in counts) and sum(1 for k in counts if "gmx_stock_kernel" in k) == 8
    inc = os.path.join(ROOT, "gmix_amd", "csrc", "gmx_stock_asm.inc")
    before = open(inc).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "gmix_amd", "csrc", "gen_stock_asm.py")],
                          stdout=subprocess.DEVNULL)
    assert open(inc).read() == before


def test_indirect_descriptions_are_validated_before_any_device_is_touched():
    """gmx_indirect_create checks its arguments on the host first: bad tables, clashing slots and
    missing next-state tables are GMX_ERR_INVALID with or without a GPU."""
    from gmix_amd._lib import IndirectDesc
    L = _lib.lib()
    tabs = (C.c_uint8 * 512)()
    h = C.c_void_p()

    def create(descs, ns=tabs, rm=tabs, streams=1):
        arr = (IndirectDesc * max(1, len(descs)))(*[IndirectDesc(*d) for d in descs])
        return L.gmx_indirect_create(C.byref(h), arr, len(descs), ns, rm, streams, 0)

    assert create([]) == -1                                   # no models
    assert create([(0, 0.02, 0, 1)]) == -1                    # empty table
    assert create([(1 << 24, 0.02, 0, 1)]) == -1              # 256*table_size+1 must fit 32 bits
    assert create([(256, 0.02, 3, 3)]) == -1                  # one slot for both predictions
    assert create([(256, 0.02, 0, 1), (256, 0.02, 1, 2)]) == -1  # two models on one slot
    assert create([(256, 0.02, -1, 1)]) == -1
    assert create([(256, 0.02, 0, 1)], ns=None) == -1
    assert create([(256, 0.02, 0, 1)], streams=0) == -1
    assert create([(256, 0.02, 2 * i, 2 * i + 1) for i in range(65)]) == -1   # more than 64 models
    if gmix_amd.device_count() == 0:
        assert create([(256, 0.02, 0, 1)]) == -4              # valid, but nothing to run it on


def test_lstm_entry_points_reject_bad_arguments():
    L = _lib.lib()
    h = C.c_void_p()
    assert L.gmx_lstm_create(C.byref(h), 0, 0) == -1          # no streams
    assert L.gmx_lstm_create(None, 1, 0) == -1
    if gmix_amd.device_count() == 0:
        assert L.gmx_lstm_create(C.byref(h), 1, 0) == -4      # no CPU fallback
    assert L.gmx_lstm_run(None, None, 1, 1, None) == -1
    assert L.gmx_lstm_forward(None, 0, 0, None, None, None) == -1
    assert L.gmx_lstm_perceive(None, 0, 3) == -1
    assert L.gmx_lstm_feed(None, None, 1, None, 1, -1, None, 0) == -1
    assert L.gmx_lstm_bank_bytes(None) == 0
    n = C.c_size_t(0)
    assert L.gmx_lstm_export(None, 0, None, C.byref(n), None, C.byref(n)) == -1
    assert L.gmx_lstm_import(None, 0, None, 0, None, 0) == -1
    assert L.gmx_lstm_copy(None, 0, None, 0) == -1
    assert L.gmx_lstm_memory_usage(None, None) == -1
    m = (C.c_uint32 * 8)(*([0xFFFF] * 8))
    assert L.gmx_lstm_set_cu_mask(None, m, 8) == -1
    assert L.gmx_indirect_set_cu_mask(None, m, 8) == -1
    assert L.gmx_group_set_cu_mask(None, m, 8) == -1
    assert L.gmx_lockstep_create(None, None, 0) == -1
    assert L.gmx_lockstep_predict(None) == -1 and L.gmx_lockstep_learn(None) == -1
    assert L.gmx_lockstep_learn_predict(None) == -1 and not L.gmx_lockstep_batch(None)
    assert L.gmx_lockstep_is_persistent(None) == 0
