"""The N > 1 path on CPU: world_size-2 gloo processes shard independent streams, run them (the
oracle stands in for the GPU bank here -- this test is about the plumbing), gather the
compressed sizes, and agree with a single process."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream_size(s):
    from gmix_amd import topology
    from oracle import gmxo
    topo = topology.synth3(24, l0=3, l1=2, table0=64, table1=8)
    pred, act, ctx, bits = gmxo.synth(24, 6, 400, seed=1000 + s, ctx_mode=1, ctx_mod=30, bit_mode=1)
    p, _ = gmxo.Bank(24, topo.skip, topo.mixers).run(pred, act, ctx, bits, want_all=False)
    return len(gmxo.encode(bits, p))


def _worker(rank, world, port, n_streams, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gmix_amd import shard
    mine = shard.local_streams(n_streams, world, rank)
    assert all(shard.stream_owner(s, world) == rank for s in mine)
    sizes = shard.gather_u64({s: _stream_size(s) for s in mine}, n_streams, dist)
    slowest = shard.max_over_ranks(1.0 + rank, dist)
    total = shard.sum_over_ranks(4096 - 1024 * rank, dist)   # a rank that had to run fewer streams
    dist.barrier()
    q.put((rank, mine, sizes.tolist(), slowest, total))
    dist.destroy_process_group()


def test_two_ranks_shard_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n_streams, port = 2, 5, _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [_stream_size(s) for s in range(n_streams)]
    owned = sorted(s for _, mine, _, _, _ in got for s in mine)
    assert owned == list(range(n_streams))          # every stream exactly once
    for rank, mine, sizes, slowest, total in got:
        assert sizes == expect                       # every rank sees every size
        assert slowest == 2.0                        # MAX over ranks
        assert total == 4096 + 3072                  # SUM over ranks


def test_shard_helpers_single_process():
    from gmix_amd import shard
    assert shard.local_streams(10, 4, 1) == [1, 5, 9]
    assert [shard.stream_owner(s, 8) for s in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert shard.gather_u64({2: 7}, 4).tolist() == [0, 0, 7, 0]
    assert shard.max_over_ranks(3.5) == 3.5
