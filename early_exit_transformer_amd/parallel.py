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


def allreduce_gradients(params, b_local: int, bucket_bytes: int = 64 << 20, group: Optional[dist.ProcessGroup] = None) -> int:
    """Data-parallel training (BASELINE config 4): after the local backward, average ``p.grad`` over the ranks, weighted by
    the ranks' utterance counts (the loss is a batch mean: global grad = sum_r grad_r * B_r / sum_r B_r).  Gradients are
    flattened into buckets of ``bucket_bytes`` in reverse parameter order (the order the backward produced them) and each
    bucket is ONE all-reduce -- two 64 MB collectives for the 31.5 M-parameter default model, sized for the per-link
    bandwidth of the xGMI ring rather than for many small launches.  Returns the number of collectives issued.
    BatchNorm statistics stay per replica."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return 0
    with_grad = [p for p in params if p.grad is not None]
    if not with_grad:
        return 0
    dev = with_grad[0].grad.device
    total = torch.tensor([float(b_local)], dtype=torch.float32, device=dev)
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    w = float(b_local) / float(total.item())
    n_coll, bucket, size = 1, [], 0

    def flush():
        nonlocal bucket, size, n_coll
        if not bucket:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bucket]).mul_(w)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
        n_coll += 1
        bucket, size = [], 0

    for p in reversed(with_grad):
        bucket.append(p)
        size += p.grad.numel() * p.grad.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    return n_coll
