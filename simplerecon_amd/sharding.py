"""Multi-GPU layout of the hot path: keyframes are independent units (reference
datasets/generic_mvs_dataset.py:602-661; test.py:257-280 keeps no cross-batch state), so the
stream is sharded round-robin over ranks -- one process per GPU -- with NO collective on the data
path.  The only exchange is the gather of the finished depth maps to rank 0 (RCCL over xGMI when
the backend is "nccl"; "gloo" in the CPU tests), once per run / per chunk, never per frame."""
from typing import List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Keyframe i goes to rank i mod world."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_items, world))


def batches(indices: List[int], batch_size: int) -> List[List[int]]:
    """Splits a rank's keyframes into batches; the last one may be ragged."""
    return [indices[i:i + batch_size] for i in range(0, len(indices), batch_size)]


def gather_results(local: torch.Tensor, n_items: int, dst: int = 0, group=None,
                   force_collective: bool = False) -> Optional[torch.Tensor]:
    """`local[j]` is the result of keyframe shard_indices(n_items, rank, world)[j].  Returns, on
    rank `dst`, the [n_items, ...] tensor in keyframe order; None elsewhere.  One collective:
    shards are padded to the longest shard so a plain gather works with ragged counts.
    `force_collective`: run the gather even in a world of one (the early return otherwise hides the RCCL
    device-buffer path from every 1-GPU test box)."""
    if not dist.is_initialized():
        if force_collective:
            raise RuntimeError("gather_results(force_collective=True) needs an initialised process group")
        return local
    if dist.get_world_size(group) == 1 and not force_collective:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_items + world - 1) // world
    if local.shape[0] != len(shard_indices(n_items, rank, world)):
        raise ValueError(f"rank {rank} holds {local.shape[0]} results, expected "
                         f"{len(shard_indices(n_items, rank, world))}")
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    local = local.contiguous()
    device = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo (the CPU-test / shared-GPU backend) has no device-tensor gather: stage through host memory.  RCCL
        # ("nccl") gathers the device buffers directly over xGMI.
        local = local.cpu()
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if bufs[0].device != device:
        bufs = [b.to(device) for b in bufs]
        local = bufs[0]
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        cnt = len(range(r, n_items, world))
        if cnt:
            out[r::world] = bufs[r][:cnt]      # keyframe i lives on rank i mod world: a strided copy, no index tensor
    return out
