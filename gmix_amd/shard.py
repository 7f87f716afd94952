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


def min_over_ranks(value, dist=None, device="cpu"):
    """MIN all-reduce of a float (the slowest rank's rate beside the fastest's)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def gather_floats(value, dist=None, device="cpu"):
    """Every rank's float, in rank order, on every rank (one SUM all-reduce of a zero-filled vector)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    import torch
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.cpu()]


def pin_to_gpu_numa_node(pci_bus_id):
    """Keep this process (and the threads it starts) on the cores of the NUMA node the GPU hangs on -- the host side
    of a rank (decay tables, record staging, and the reference's feature models when whole files are compressed)
    belongs next to its device.  Returns the cpus kept, or None when sysfs does not say."""
    import os
    try:
        node = int(open(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        keep = cpus & os.sched_getaffinity(0)
        if not keep:
            return None
        os.sched_setaffinity(0, keep)
        return sorted(keep)
    except (OSError, ValueError):
        return None
