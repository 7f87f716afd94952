"""The drop-in's HOST logic against the real reference, without a GPU: the reference's tester built
with Predictor::AddMixers constructing gmx::GpuMixer (gmix_amd/host/gmx_model_adapter.h), the eight
C-ABI calls the adapter makes answered by the oracle (tests/cpp/gmx_abi_oracle_shim.c -- test-only;
the product library has no CPU path), beside the stock build: registration with ShortTermMemory /
LongTermMemory, one bank call for 33 Predict/Learn calls, staging through LongTermMemory::mixers for
the reference's own checkpoint writers and readers, Copy.  The same comparison against libgmxmix.so
on an MI355X is tests/test_gpu_dropin.py."""
import os

import pytest

from dropin_common import REF, compare, run_pair


@pytest.mark.slow
def test_reference_tester_with_adapter_equals_stock(tmp_path):
    for exe in ("ref_tester_strict", "ref_tester_shim"):
        if not os.path.exists(os.path.join(REF, exe)):
            pytest.skip(f"oracle/_ref/{exe} not built (needs /root/reference: make -C oracle/ref_build dropin)")
    da, db = run_pair("ref_tester_strict", "ref_tester_shim", 2500, 300, tmp_path)
    compare(da, db)


@pytest.mark.slow
def test_reference_tester_with_the_whole_device_chain_adapters_equals_stock(tmp_path):
    """... and with LstmModel -> gmx::GpuLstmModel and the 41 Indirect -> gmx::GpuIndirect as well (their
    C-ABI calls answered by the oracle's restatements, tests/cpp/gmx_abi_oracle_shim2.c): the bank that runs
    when the LAST Indirect model is called, active_models back in index order, the .long sections of both
    written by the reference's own serialiser from what the adapters staged.  Without TestGeneration: it
    checkpoints after a Predict whose byte is never perceived, which the device LSTM bank refuses."""
    for exe in ("ref_tester_strict", "ref_tester_chain_shim"):
        if not os.path.exists(os.path.join(REF, exe)):
            pytest.skip(f"oracle/_ref/{exe} not built (needs /root/reference: make -C oracle/ref_build dropin)")
    da, db = run_pair("ref_tester_strict", "ref_tester_chain_shim", 1200, 0, tmp_path)
    compare(da, db, generation=False)
