"""The drop-in's HOST logic against the real reference, without a GPU: the reference's tester built
with Predictor::AddMixers constructing gmx::GpuMixer (gmix_amd/host/gmx_model_adapter.h), the eight
C-ABI calls the adapter makes answered by the oracle (tests/cpp/gmx_abi_oracle_shim.c -- test-only;
the product library has no CPU path), beside the stock build: registration with ShortTermMemory /
LongTermMemory, one bank call for 33 Predict/Learn calls, staging through LongTermMemory::mixers for
the reference's own checkpoint writers and readers, Copy.  The same comparison against libgmxmix.so
on an MI355X is tests/test_gpu_dropin.py."""
import os

import pytest

from dropin_common import compare, exe as exe_path, run_all


@pytest.mark.slow
def test_reference_tester_with_adapters_equals_stock(tmp_path):
    """Three builds of the reference's tester at once on the same bytes: stock; mixers through
    gmx::GpuMixer; and the whole device chain -- LstmModel -> gmx::GpuLstmModel and the 41 Indirect ->
    gmx::GpuIndirect as well (the bank that runs when the LAST Indirect model is called, active_models back
    in index order, the .long sections written by the reference's own serialiser from what the adapters
    staged).  TestGeneration included: it checkpoints after a Predict whose byte is never perceived, i.e. the
    LSTM bank between its forward and its perceive (include/gmxmix.h)."""
    exes = ("ref_tester_strict", "ref_tester_shim", "ref_tester_chain_shim")
    for exe in exes:
        if not os.path.exists(exe_path(exe)):
            pytest.skip(f"{exe_path(exe)} not built (needs /root/reference: make -C oracle/ref_build full && make -C dropin)")
    stock, mixers, chain = run_all([(exes[0], 300), (exes[1], 300), (exes[2], 300)], 1500, tmp_path)
    compare(stock, mixers)
    compare(stock, chain)
