"""Multi-GPU multi-start plumbing (north-star config 4): one process per GPU, restarts sharded by rank, and
one RCCL min-all-reduce of an 8-byte key per round.  No counterpart in the reference.

key = (f32 cost bits << 32) | restart id — order-preserving for cost >= 0, ties go to the lowest restart id,
identical to tl_pack_cost_key in the C ABI.  The collective runs on whatever backend the process group was
created with: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np
import torch


def shard(rank, restarts_per_rank):
    """Weak scaling: rank k owns restarts [k*R, (k+1)*R)."""
    first = rank * restarts_per_rank
    return first, restarts_per_rank


def pack_keys(costs, first):
    """costs: float32 tensor [R] (any device) of restarts first..first+R -> int64 keys [R]."""
    ids = torch.arange(first, first + costs.numel(), dtype=torch.int64, device=costs.device)
    return (costs.contiguous().view(torch.int32).to(torch.int64) << 32) | ids


def unpack_key(key):
    key = int(key)
    cost = float(np.frombuffer(np.uint32((key >> 32) & 0xFFFFFFFF).tobytes(), dtype=np.float32)[0])
    return cost, key & 0xFFFFFFFF


def allreduce_best(local_keys, dist=None):
    """min over the local keys, then min-all-reduce across ranks; returns a 1-element int64 tensor."""
    best = local_keys.min().reshape(1)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(best, op=dist.ReduceOp.MIN)
    return best


def aggregate_throughput(candidates_local, seconds_local, device, dist=None):
    """whole-job candidates (sum over ranks) and the slowest rank's time (max over ranks)."""
    c = torch.tensor([int(candidates_local)], dtype=torch.int64, device=device)
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(c.item()), float(t.item())
