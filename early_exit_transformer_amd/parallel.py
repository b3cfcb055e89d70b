"""Utterance-batch sharding over the GPUs of one node (one process per GPU, RCCL via
torch.distributed backend "nccl"; "gloo" in the CPU tests).

The encoder forward has no exchange step: utterances are independent in eval mode,
so each rank runs the whole stack on its slice with replicated weights.  The only collective is
the reduction of the summed per-exit CTC loss (reference train.py:53-65, ``reduction='mean'`` =
batch mean): rank-local batch means are combined as sum(mean_r * B_r) / sum(B_r)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n utterances owned by ``rank``; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def combine_exit_losses(local_mean: torch.Tensor, b_local: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All-reduce of per-exit batch-mean losses [E] (or a scalar) to the global-batch mean.
    One collective of E+1 floats; latency-bound over xGMI."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return local_mean  # single shard: the local batch mean already is the global one
    buf = torch.cat([local_mean.reshape(-1).to(torch.float32) * float(b_local),
                     torch.tensor([float(b_local)], dtype=torch.float32, device=local_mean.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return (buf[:-1] / buf[-1]).reshape(local_mean.shape)
