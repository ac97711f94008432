"""Multi-GPU sharding of the hot path: one process per GPU, windows (scenes) block-sharded.

Social pooling never crosses a window (zero-comm alternative), so the
data path needs no collective; the only communication is an optional all_gather of the finished
trajectories/scores.  Works with backend "nccl" (= RCCL over xGMI on ROCm) and with "gloo" on
CPU tensors (used by tests/test_dist_gloo.py, world_size 2)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_windows(n_windows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of windows owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world) or n_windows < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_windows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_windows(len(batch), rank, world)
    return list(batch[lo:hi])


def gather_results(local, n_windows: int, group=None):
    """All-gather per-window results (first dim = local windows) into the global window order.
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_windows(n_windows, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError("local shard has %d windows, expected %d" % (local.shape[0], hi - lo))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: h - l] for o, (l, h) in zip(out, sizes)], dim=0)


def allreduce_mean_(flat, group=None):
    """In-place mean of ONE flat gradient tensor over the data-parallel ranks (no-op when torch.distributed is not
    initialised or world_size == 1).  Each rank's loss is a mean over ITS present agents, so this is the usual
    data-parallel estimate of the global-batch gradient.  RCCL over xGMI on GPU tensors, gloo on CPU tensors."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    return flat
