"""Multi-GPU multi-start plumbing (north-star config 4): one process per GPU, restarts sharded by rank, and
one RCCL min-all-reduce of an 8-byte key per round, followed by the winning tour (n x 4 B) to every rank.  No counterpart
in the reference.

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


def shard_total(rank, world, total):
    """Strong scaling (BASELINE configs[3]: "256 random restarts sharded 1/2/4/8 GPUs"): `total` restarts in all, rank k
    owns the k-th contiguous block; the first total % world ranks take one restart more.  Returns (first, count)."""
    base, extra = divmod(int(total), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


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
    if dist is not None and dist.is_initialized():  # also with one rank (bench.py's TL_BENCH_FORCE_DIST rehearsal of the RCCL path)
        dist.all_reduce(best, op=dist.ReduceOp.MIN)
    return best


def share_best_tour(local_keys, local_tours, best_key, dist=None):
    """The tour behind `best_key` on every rank, without a host round trip: the rank whose local minimum IS the global key
    contributes that tour, everybody else zeros, and one SUM-all-reduce of n x 4 B (40 KB at n = 10^4) delivers it —
    restart ids are unique across ranks, so exactly one rank matches.  (all-reduce the key, then hand the
    owner's tour round; a broadcast would need the owner's rank on the host first.)
    local_keys int64 [R], local_tours int32 [R, n], best_key int64 [1] from allreduce_best."""
    idx = local_keys.argmin()
    tour = local_tours[idx]
    if dist is None or not dist.is_initialized():
        return tour.clone()
    mine = (local_keys[idx] == best_key[0])
    buf = torch.where(mine, tour, torch.zeros_like(tour))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def aggregate_throughput(candidates_local, seconds_local, device, dist=None):
    """whole-job candidates (sum over ranks) and the slowest rank's time (max over ranks)."""
    c = torch.tensor([int(candidates_local)], dtype=torch.int64, device=device)
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(c.item()), float(t.item())
