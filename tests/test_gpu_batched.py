"""The run-ahead compressor on an MI355X (BASELINE.json configs[2] / [3]): the reference's own
88 feature models and coder on the host cores, running ahead of the 33 mixers, which libgmxmix.so takes in
double-buffered batches (gmix_amd/host/gmx_batched.h).  Built by dropin/Makefile from the
reference's sources where they lie, with Predictor::AddMixers constructing gmx::GpuMixer and RunCompression calling
gmx::BatchedCompress: the reference calls the product.  Every file must equal the stock build's."""
import os
from concurrent.futures import ThreadPoolExecutor

import pytest

from batched_common import compress_pair, corpus, gmix, need, run_many, same_outputs
from dropin_common import checkpoint_after_batches, compare, run_all, run_pair, same_checkpoint

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exe", ["gmix_batched", "gmix_chain_batched"])
def test_batched_compress_100k_equals_stock_and_round_trips(gpu, tmp_path, exe):
    """`gmix -c` of 100 000 bytes of text: 391 chunks of 2 048 bits through the ring of batches.  Same compressed
    bytes as the stock build, same analysis tables (1 000 rows whose final-mixer entropy is computed from the
    returned chunks), and the STOCK build's `gmix -d` restores the input from the run-ahead compressor's file.
    gmix_chain_batched: LSTM -> 41 Indirect models -> 33 mixers on the device, each chunk a chain of three batched
    kernels with lstm_prediction_context and 83 of the mixers' 90 inputs handed on inside HBM; the host runs PPMd,
    the match models, the context hashes and the coder; the LSTM's and the Indirect models' analysis columns come
    from the device too."""
    need("gmix_strict", exe)
    src, stock, batched = compress_pair("gmix_strict", exe, corpus(100000), tmp_path)
    same_outputs(stock, batched)
    gmix("gmix_strict", "-d", batched / "c", stock / "back", stock)
    assert (stock / "back").read_bytes() == src.read_bytes()


def test_reference_tester_chain_with_batched_compression_equals_stock(gpu, tmp_path):
    """The same with the whole device chain."""
    need("ref_tester_strict", "ref_tester_chain_batched")
    da, db = run_pair("ref_tester_strict", "ref_tester_chain_batched", 6000, 400, tmp_path)
    compare(da, db)


def test_reference_tester_with_batched_compression_equals_stock(gpu, tmp_path):
    """The reference's five tests with RunCompression running ahead on the device: TestCompression is batched, the
    restart / Copy / decode / generation tests go per bit (sessions) on banks that batches have been through; the
    tester compares their files with the batched one itself, and all it leaves equals the stock build's."""
    need("ref_tester_strict", "ref_tester_batched")
    da, db = run_pair("ref_tester_strict", "ref_tester_batched", 30000, 2000, tmp_path)
    compare(da, db)


@pytest.mark.parametrize("exe", ["gmix_many", "gmix_chain_many"])
def test_64_files_side_by_side_equal_stock(gpu, tmp_path, exe):
    """64 (mixers only: 32) Predictors on as many host threads, their mixers streams of ONE gmx_group (gmix_chain_many: and their LSTMs
    and Indirect models 64 streams of one gmx_lstm / gmx_indirect), one launch per bank and 2 048-bit chunk for all of
    them; files of 12 000 .. 18 300 bytes (3 000 .. 9 300 with the mixers alone on the device) starting at different
    places of the corpus, so they end in different rounds.  Every output is the stock build's `gmix -c` of the same file."""
    need("gmix_strict", exe)
    base = 12000 if exe == "gmix_chain_many" else 3000   # (mixers only: the host's 88 feature models make it 9 us per bit)
    n = 64 if exe == "gmix_chain_many" else 32   # (mixers only: the Predictors are built one after the other)
    files = []
    for k in range(n):
        f = tmp_path / f"f{k}"
        f.write_bytes(corpus(base + 100 * k, 1531 * k))
        files.append(f)
    st = run_many(exe, files, tmp_path / "out", 2048)
    assert st["failed"] == 0 and st["files"] == n
    assert st["device_bits"] == 8 * sum(base + 100 * k for k in range(n))

    def stock(k):
        d = tmp_path / f"s{k}"
        d.mkdir()
        gmix("gmix_strict", "-c", files[k], d / "c", d)
        return (d / "c").read_bytes()

    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(stock, range(n)))
    for k in range(n):
        assert refs[k] == (tmp_path / "out" / f"{k}.gmix").read_bytes(), f"file {k} differs from gmix_strict -c"
    print(f"{exe}, {n} files: {st['bits_per_second']:.3g} bits/s aggregate, {st['wall_seconds']:.2f} s, "
          f"{st['launches']} launches, {st['pinned_threads']} threads pinned")


@pytest.mark.parametrize("exe,chunk", [("gmix_many", 8), ("gmix_many", 1000), ("gmix_chain_many", 8),
                                       ("gmix_chain_many", 1000)])
def test_small_chunks_and_ragged_ends(gpu, tmp_path, exe, chunk):
    """Chunks of one byte (every launch is a ragged one at the end) and of 1 000 bits for files of 1 .. 2 000 bytes."""
    need("gmix_strict", exe)
    sizes = (1, 300, 2000, 2000, 777)
    files = []
    for k, n in enumerate(sizes):
        f = tmp_path / f"f{k}"
        f.write_bytes(corpus(n, 5000 * k))
        files.append(f)
    st = run_many(exe, files, tmp_path / "out", chunk)
    assert st["failed"] == 0 and st["device_bits"] == 8 * sum(sizes)
    for k, f in enumerate(files):
        gmix("gmix_strict", "-c", f, tmp_path / f"ref{k}", tmp_path)
        assert (tmp_path / f"ref{k}").read_bytes() == (tmp_path / "out" / f"{k}.gmix").read_bytes(), f"file {k}"


def test_state_left_behind_equals_the_per_bit_loop(gpu, tmp_path):
    """What a run-ahead compression LEAVES on the device and on the blackboard: Predictor::WriteCheckpoint straight
    after gmx::BatchedCompressor over 8 001 bytes (31 chunks of 2 048 bits and a ragged one; chain: 65 of 1 000)
    equals the checkpoint the stock tester writes after the same bytes through its per-bit loop
    (tester.cpp:32-59): the three banks' state as the reference's serialisers write it, the mixers' outputs, the
    Indirect models' and the LSTM's prediction slots (gmx_indirect_slots_get) and lstm_prediction_context."""
    need("ref_tester_strict", "gmix_batched_ckpt", "gmix_chain_batched_ckpt")
    (stock,) = run_all([("ref_tester_strict", 0)], 16000, tmp_path)
    for exe, chunk in (("gmix_batched_ckpt", 2048), ("gmix_chain_batched_ckpt", 1000)):
        same_checkpoint(os.path.join(stock, "restart"), checkpoint_after_batches(exe, stock, 16000, chunk, tmp_path))


@pytest.mark.parametrize("exe,n_files,base,groups", [("gmix_chain_many", 64, 1500, 1), ("gmix_many", 8, 600, 1),
                                                     ("gmix_chain_many", 12, 400, 2), ("gmix_chain_many", 12, 400, 0)])
def test_files_restored_side_by_side_in_lock_step(gpu, tmp_path, exe, n_files, base, groups):
    """gmx::BatchedDecompressFiles on the device: the reference's own Decoder (coder/decoder.cpp:19-39) per file, each on
    a fibre of a few worker threads; every coded bit of all files is ONE gmx_chainstep step -- LSTM, 41 Indirect models and
    33 mixers of all streams in one hipGraph (gmix_many: the mixers alone).  Files the run-ahead compressor wrote AND
    files the stock build wrote (`gmix_strict -c`) are restored byte for byte; lengths differ, so streams leave one by
    one.  groups = 2: the files in two pools, each a lock step of its own, taken in turn by the same worker threads --
    gmx_chainstep_launch of one, the other's fibres, gmx_chainstep_wait (their banks brought up one thread at a time: a
    synchronous copy beside another thread's graph capture fails both).  groups = 0 here: one pool, and GMX_CS_NO_BAR -- the
    step's inputs fetched by its first kernel, as on a host whose device memory the CPU cannot store into."""
    env = {"GMX_CS_NO_BAR": "1"} if groups == 0 else None
    groups = max(groups, 1)
    need("gmix_strict", exe)
    files = []
    for k in range(n_files):
        f = tmp_path / f"f{k}"
        f.write_bytes(corpus(base + 37 * k, 1531 * k))
        files.append(f)
    st = run_many(exe, files, tmp_path / "c", 2048)
    assert st["failed"] == 0
    coded = [tmp_path / "c" / f"{k}.gmix" for k in range(n_files)]
    for k in (0, n_files // 2, n_files - 1):   # ... and the stock build's own files among them
        gmix("gmix_strict", "-c", files[k], tmp_path / f"stock{k}", tmp_path)
        assert (tmp_path / f"stock{k}").read_bytes() == coded[k].read_bytes()
        coded[k] = tmp_path / f"stock{k}"
    st = run_many(exe, coded, tmp_path / "back", 2048, extra=("-d", "--groups", str(groups)), env=env)
    assert st["mode"] == "decompress" and st["failed"] == 0
    for k, f in enumerate(files):
        assert (tmp_path / "back" / f"{k}.out").read_bytes() == f.read_bytes(), f"file {k}"
    print(f"{exe} -d, {n_files} files: {st['bits_per_second']:.3g} bits/s aggregate over {st['wall_seconds']:.2f} s, "
          f"{st['launches']} steps = {st['wall_seconds'] / st['launches'] * 1e6:.1f} us per step")
