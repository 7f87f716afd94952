"""Rank program of tests/test_gpu_shard_nccl.py (started by torch.distributed.run, one rank per
visible GPU): every rank runs ITS streams through the HIP mixer bank on its own GPU, codes them
with the (oracle's) arithmetic coder on the host, and the compressed sizes are gathered over
RCCL from device tensors -- BASELINE configs[4] in small."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n_streams, out_path = int(sys.argv[1]), sys.argv[2]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import numpy as np

    import gmix_amd
    from gmix_amd import shard, topology
    from oracle import gmxo  # the coder and the stream generator: test infrastructure
    topo = topology.stock(90)
    mine = shard.local_streams(n_streams, world, rank)
    T = 256
    sizes = {}
    if mine:
        g = gmix_amd.MixerGroup(topo, len(mine), device=local_rank)
        b = gmix_amd.Batch(g, T)
        recs = []
        for i, s in enumerate(mine):
            pred, act, ctx, bits = gmxo.synth(90, 33, T, seed=500 + s, ctx_mode=3, ctx_mod=6, zero_mod=8, bit_mode=1)
            b.set_records(i, pred, act, ctx, bits)
            recs.append(bits)
        b.upload()
        g.run(b)
        b.download()
        b.wait()
        for i, s in enumerate(mine):
            sizes[s] = len(gmxo.encode(recs[i], np.array(b.p[i])))
        b.close()
        g.close()
    full = shard.gather_u64(sizes, n_streams, dist, device=f"cuda:{local_rank}")
    slowest = shard.max_over_ranks(1.0 + rank, dist, device=f"cuda:{local_rank}")
    total = shard.sum_over_ranks(len(mine), dist, device=f"cuda:{local_rank}")
    dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        json.dump({"world": world, "backend": dist.get_backend(), "sizes": full.tolist(), "slowest": slowest,
                   "total": total}, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
