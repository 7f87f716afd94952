"""The run-ahead compressor's HOST logic without a GPU (gmix_amd/host/gmx_batched.h + the run-ahead mode and
MixerPool of gmx_model_adapter.h): the reference's CLI, tester and a many-file driver built on it, the C-ABI calls
answered by the oracle (tests/cpp/gmx_abi_oracle_shim.c -- test-only; the product has no CPU path), beside the
stock build.  What is checked here: records written at the right bit, chunks handed in and drained in order
through the ring of batches, ragged ends, the coder fed in order, the analysis rows, thread hand-offs.  The
same comparisons against libgmxmix.so on an MI355X: tests/test_gpu_batched.py."""
import os

import pytest

from batched_common import compress_pair, corpus, gmix, need, run_many, same_outputs
from dropin_common import checkpoint_after_batches, compare, exe as exe_path, run_all, same_checkpoint


def _skip_unless(*exes):
    for exe in exes:
        if not os.path.exists(exe_path(exe)):
            pytest.skip(f"{exe_path(exe)} not built (needs /root/reference: make -C oracle/ref_build full && make -C dropin)")


@pytest.mark.parametrize("exe", ["gmix_batched_shim", "gmix_chain_batched_shim"])
def test_batched_cli_writes_the_stock_file_and_tables(tmp_path, exe):
    """5 000 bytes: 19+ chunks of 2 048 bits and a ragged last one; analysis on (40 bits per row: 1 000 rows whose
    final-mixer column comes from the returned chunks).  `chain`: the LSTM and the 41 Indirect models record as
    well (PPM byte distributions, contexts) and run in batches in front of the mixers, lstm_prediction_context
    handed on between the batches; the LSTM's and the 30 analysed Indirect predictions' table columns come from
    the returned chunks too."""
    _skip_unless("gmix_strict", exe)
    src, stock, batched = compress_pair("gmix_strict", exe, corpus(5000, 777), tmp_path)
    same_outputs(stock, batched)
    gmix("gmix_strict", "-d", batched / "c", stock / "back", stock)
    assert (stock / "back").read_bytes() == src.read_bytes()


@pytest.mark.parametrize("exe,chunk", [("gmix_many_shim", 8), ("gmix_many_shim", 72), ("gmix_many_shim", 4096),
                                       ("gmix_chain_many_shim", 8), ("gmix_chain_many_shim", 1000)])
def test_many_files_ragged_lengths(tmp_path, exe, chunk):
    """Three Predictors on three threads share one group; files of 1 / 613 / 1 500 bytes end in different rounds
    (a stream that has left must not hold the others up), chunks of one byte up to more than the longest file."""
    _skip_unless("gmix_strict", exe)
    files = []
    for k, n in enumerate((1, 613, 1500)):
        f = tmp_path / f"f{k}"
        f.write_bytes(corpus(n, 4000 * k))
        files.append(f)
    st = run_many(exe, files, tmp_path / "out", chunk)
    assert st["failed"] == 0 and st["device_bits"] == 8 * (1 + 613 + 1500)
    for k, f in enumerate(files):
        gmix("gmix_strict", "-c", f, tmp_path / f"ref{k}", tmp_path)
        assert (tmp_path / f"ref{k}").read_bytes() == (tmp_path / "out" / f"{k}.gmix").read_bytes(), f"file {k}"


@pytest.mark.slow
def test_reference_tester_with_batched_compression(tmp_path):
    """The reference's tester with RunCompression running ahead: TestCompression goes through
    gmx::BatchedCompress, the restart / Copy / decode tests through the per-bit path on the bank the batch
    left behind -- the tester itself compares their files with the batched one, and everything it leaves equals
    the stock build's."""
    _skip_unless("ref_tester_strict", "ref_tester_batched_shim", "ref_tester_chain_batched_shim")
    stock, batched, chain = run_all([("ref_tester_strict", 200), ("ref_tester_batched_shim", 200),
                                     ("ref_tester_chain_batched_shim", 200)], 800, tmp_path)
    compare(stock, batched)
    compare(stock, chain)


@pytest.mark.parametrize("exe,chunk", [("gmix_batched_ckpt_shim", 2048), ("gmix_chain_batched_ckpt_shim", 1000)])
def test_state_left_behind_equals_the_per_bit_loop(tmp_path, exe, chunk):
    """What a run-ahead compression LEAVES: Predictor::WriteCheckpoint straight after gmx::BatchedCompressor over
    401 bytes equals the checkpoint the reference's tester writes after the same 401 bytes through its
    Predict/Encode/Perceive/Learn loop (tester.cpp:32-59) -- the banks' state as the reference's serialisers write
    it, the blackboard (mixer outputs, final output, lstm_prediction_context from the device in `chain`), the
    LSTM's byte-range state, every host model."""
    _skip_unless("ref_tester_strict", exe)
    (stock,) = run_all([("ref_tester_strict", 0)], 800, tmp_path)
    ck = checkpoint_after_batches(exe, stock, 800, chunk, tmp_path)
    same_checkpoint(os.path.join(stock, "restart"), ck)


@pytest.mark.parametrize("exe,side_by_side", [("gmix_chain_many_shim", True), ("gmix_many_shim", False)])
def test_predictors_built_side_by_side_where_no_constructor_draws(tmp_path, exe, side_by_side):
    """Four files.  With the LSTM on the device its initial weights -- a constant: every Predictor constructor begins
    with srand(0xDEADBEEF), predictor.cpp:18 -- come from the pool's one draw (MixerPool::DrawLstmInit) and the
    Predictors behind the first are constructed at once on their threads; with the host's own LstmModel drawing from
    rand() (lstm-layer.h:41) constructions stay serial.  Either way every file is the stock build's."""
    _skip_unless("gmix_strict", exe)
    files = []
    for k in range(4):
        f = tmp_path / f"f{k}"
        f.write_bytes(corpus(110 + 31 * k, 2500 * k))
        files.append(f)
    st = run_many(exe, files, tmp_path / "out", 256)
    assert st["failed"] == 0 and st["parallel_construction"] is side_by_side
    assert st["total_seconds"] >= st["build_seconds"] + st["wall_seconds"] - 2e-3   # (build_seconds is printed to the millisecond)
    for k, f in enumerate(files):
        gmix("gmix_strict", "-c", f, tmp_path / f"ref{k}", tmp_path)
        assert (tmp_path / f"ref{k}").read_bytes() == (tmp_path / "out" / f"{k}.gmix").read_bytes(), f"file {k}"


@pytest.mark.parametrize("exe,cpus,groups", [("gmix_many_shim", 2, 1), ("gmix_chain_many_shim", 3, 1),
                                             ("gmix_chain_many_shim", 1, 1), ("gmix_chain_many_shim", 4, 2),
                                             ("gmix_many_shim", 3, 3)])
def test_many_files_restored_in_lock_step(tmp_path, exe, cpus, groups):
    """gmx::BatchedDecompressFiles: five files the STOCK build compressed (0 / 1 / 613 / 300 / 613 bytes) restored
    together -- the reference's own Decoder per file, each on a fibre, `cpus` worker threads, one gmx_chainstep step per
    coded bit for all of them; files that end early sit the remaining steps out, every file's last Learn is a step
    without a Predict.  `groups` > 1: the files in that many pools, each a lock step of its own, taken in turn by the same
    workers, who also take over each other's fibres (Predictors that draw from rand() -- the mixers-only build -- still
    constructed one at a time).
    Byte-identical to the inputs."""
    _skip_unless("gmix_strict", exe)
    files, coded = [], []
    for k, n in enumerate((0, 1, 613, 300, 613)):
        f = tmp_path / f"f{k}"
        f.write_bytes(corpus(n, 3000 * k))
        files.append(f)
        gmix("gmix_strict", "-c", f, tmp_path / f"c{k}", tmp_path)
        coded.append(tmp_path / f"c{k}")
    st = run_many(exe, coded, tmp_path / "back", 2048, extra=("-d", "--cpus", str(cpus), "--groups", str(groups)))
    assert st["mode"] == "decompress" and st["failed"] == 0 and st["launches"] >= 8 * 613 + 1   # (summed over the groups)
    for k, f in enumerate(files):
        assert (tmp_path / "back" / f"{k}.out").read_bytes() == f.read_bytes(), f"file {k}"
