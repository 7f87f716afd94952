"""Multi-GPU layout of the mixer path: independent byte-streams shard across the GPUs of a
node, one process per GPU.  Nothing crosses GPUs on the data path --
bit t+1 of a stream depends on bit t's update, so a single stream never splits -- and the only
collectives are a barrier, a MAX over the ranks' elapsed times and one gather of per-stream
results (compressed sizes) over RCCL (torch.distributed backend "nccl"; "gloo" in CPU tests)."""
import numpy as np


def stream_owner(stream, world):
    """stream s -> rank s mod world."""
    return stream % world


def local_streams(n_streams, world, rank):
    """Global stream ids this rank owns, ascending."""
    return list(range(rank, n_streams, world))


def gather_u64(local, n_streams, dist=None, device="cpu"):
    """local: {global stream id: value}.  Returns the full int64 array on every rank (one tiny
    SUM all-reduce of a zero-filled vector: latency-bound, link bandwidth irrelevant)."""
    full = np.zeros(n_streams, np.int64)
    for s, v in local.items():
        full[s] = v
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return full
    import torch
    t = torch.from_numpy(full).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX all-reduce of a float (the benchmark's elapsed time)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device="cpu"):
    """SUM all-reduce of an integer (streams per rank can differ if a rank had less free HBM)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
