"""Utterance-batch sharding over the GPUs of one node (one process per GPU, RCCL via
torch.distributed backend "nccl"; "gloo" in the CPU tests).

The encoder forward has no exchange step: utterances are independent in eval mode,
so each rank runs the whole stack on its slice with replicated weights.  The only collective of the
forward path is the reduction of the summed per-exit CTC loss (reference train.py:53-65,
``reduction='mean'`` = batch mean): rank-local batch means are combined as sum(mean_r * B_r) / sum(B_r).

Training (BASELINE.json configs[3]) adds the gradient all-reduce.  ``GradBuckets`` keeps every gradient as a VIEW of a few
flat fp32 buffers -- one bucket per exit group, in the order the backward finishes them -- so a bucket is all-reduced in
place (no ``cat``, no copy back), and ``eec_train_backward_ex`` reports each finished exit group to the host, so bucket e's
collective runs on RCCL's stream under the backward of group e - 1.  Nothing on the per-step path builds a tensor from host
data or reads one back (no ``torch.tensor(...)``, no ``.item()``): the shard sizes are exchanged once and cached."""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n utterances owned by ``rank``; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# Rehearsal switch (EEC_DP_SINGLE_RANK=1, or set the attribute): a process group of ONE rank counts as active, so that on a
# one-GPU box every collective of the path still goes through the backend -- RCCL's stream hand-over, its work objects, the
# bucket bookkeeping -- with nothing to exchange.  Results are those of the single-GPU step (weight 1, sum over one rank).
SINGLE_RANK_COLLECTIVES = os.environ.get("EEC_DP_SINGLE_RANK") == "1"


def _active(group) -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or SINGLE_RANK_COLLECTIVES


# (device, b_local) -> device-resident [b_local] fp32: built once, so a step never pays a blocking pageable host -> device
# copy (the stall eec_upload_i64 removed for `lengths`)
_count_cache: Dict[Tuple[torch.device, int], torch.Tensor] = {}


def _count_tensor(b_local: int, device: torch.device) -> torch.Tensor:
    key = (device, int(b_local))
    t = _count_cache.get(key)
    if t is None:
        t = torch.full((1,), float(b_local), dtype=torch.float32, device=device)
        _count_cache[key] = t
    return t


def shard_weight(b_local: int, device: torch.device, group: Optional[dist.ProcessGroup] = None) -> float:
    """b_local / (global batch): this rank's weight in a batch mean, as a python float.  SET-UP TIME ONLY
    (``enable_data_parallel``, where the shard size is fixed by contract): one all-reduce and one read-back on EVERY call --
    never cached, so every rank that calls it issues the collective (a cache keyed on this rank's size alone would let a rank
    whose size did not change skip a collective its peers enter).  The per-step path uses ``shard_weight_tensor``."""
    if not _active(group):
        return 1.0
    return float(shard_weight_tensor(b_local, device, group).item())


def shard_weight_tensor(b_local: int, device: torch.device, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """The same weight as a device-resident [1] fp32 tensor: ONE all-reduce of one float, issued unconditionally by every
    rank on every call, no host read-back -- safe on a per-step path whose shard sizes change between steps (a last partial
    batch)."""
    cnt = _count_tensor(b_local, device)
    if not _active(group):
        return torch.ones_like(cnt)
    total = cnt.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return cnt / total


def combine_exit_losses(local_mean: torch.Tensor, b_local: int, group: Optional[dist.ProcessGroup] = None,
                        equal_shards: bool = False) -> torch.Tensor:
    """All-reduce of per-exit batch-mean losses [E] (or a scalar) to the global-batch mean: ONE collective of E + 1 floats
    (E with ``equal_shards``: every rank holds the same number of utterances, so the global mean is the mean of the local
    ones and no count travels); latency-bound over xGMI.  No host -> device copy and no read-back on this path."""
    if not _active(group):
        return local_mean  # single shard: the local batch mean already is the global one
    flat = local_mean.reshape(-1).to(torch.float32)
    if equal_shards:
        buf = flat / float(dist.get_world_size(group))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return buf.reshape(local_mean.shape)
    cnt = _count_tensor(b_local, local_mean.device)
    buf = torch.cat([flat * cnt, cnt])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return (buf[:-1] / buf[-1]).reshape(local_mean.shape)


class GradBuckets:
    """Flat gradient storage for data-parallel training (BASELINE config 4; reference side train.py:53-70: the loss is a batch
    mean, so the global gradient is sum_r grad_r * B_r / sum_r B_r).

    ``names_params``: the model's (name, parameter) pairs.  ``group_of(name)`` maps a parameter to the exit group whose
    backward produces it last (None: the stem, finished at the very end); the default understands the Early_conformer /
    full_conformer names (``conformer.e.*``, ``linears.e.*`` / ``linears_1.e.*``).  One bucket per exit group + one for the
    rest; buckets smaller than ``min_bucket_bytes`` are merged into the next one to finish.  ``view(name)`` is what the
    backward writes a gradient into; autograd then installs exactly that tensor as ``p.grad`` (``zero_grad(set_to_none=True)``,
    the torch default the reference's ``optimizer.zero_grad()`` gets).  ``allreduce_bucket`` / ``allreduce_all`` scale by the
    shard weight and all-reduce the flat buffer in place when every ``p.grad`` of the bucket still is its view; a bucket whose
    gradients live elsewhere (accumulated into older tensors, set by the user) takes the gather / scatter path."""

    def __init__(self, names_params: Sequence[Tuple[str, torch.nn.Parameter]], n_groups: int, group_of=None,
                 min_bucket_bytes: int = 4 << 20):
        self.n_groups = int(n_groups)
        group_of = group_of or self._default_group_of
        per: Dict[int, List[Tuple[str, torch.nn.Parameter]]] = {}
        for n, p in names_params:
            if not p.requires_grad:
                continue
            g = group_of(n)
            per.setdefault(-1 if g is None else int(g), []).append((n, p))
        # buckets in the order the backward completes them: group E-1, ..., 0, then the stem (-1)
        order = [g for g in range(self.n_groups - 1, -1, -1) if g in per] + ([-1] if -1 in per else [])
        merged: List[Tuple[List[int], List[Tuple[str, torch.nn.Parameter]]]] = []
        carry_g: List[int] = []
        carry_p: List[Tuple[str, torch.nn.Parameter]] = []
        for g in order:
            carry_g.append(g)
            carry_p += per[g]
            if sum(p.numel() * 4 for _, p in carry_p) >= min_bucket_bytes or g == order[-1]:
                merged.append((carry_g, carry_p))
                carry_g, carry_p = [], []
        self.buckets: List[dict] = []
        self._where: Dict[str, Tuple[int, int, int]] = {}
        for groups, plist in merged:
            dev = plist[0][1].device
            n_el = sum(p.numel() for _, p in plist)
            flat = torch.zeros(n_el, dtype=torch.float32, device=dev)
            off = 0
            for n, p in plist:
                if p.dtype != torch.float32 or p.device != dev:
                    raise ValueError(f"parameter {n}: flat gradient buckets hold fp32 parameters of one device")
                self._where[n] = (len(self.buckets), off, p.numel())
                off += p.numel()
            # the bucket is complete once the backward has finished the LAST (lowest) of its groups
            self.buckets.append({"flat": flat, "params": plist, "ready_after": min(groups), "work": None})
        self._pending: List = []

    @staticmethod
    def _default_group_of(name: str):
        parts = name.split(".")
        if parts[0] in ("conformer", "linears", "linears_1") and len(parts) > 1 and parts[1].isdigit():
            return int(parts[1])
        return None

    def view(self, name: str, like: torch.Tensor) -> Optional[torch.Tensor]:
        """The slice of its bucket that holds the gradient of ``name`` (shaped like the parameter), or None."""
        loc = self._where.get(name)
        if loc is None:
            return None
        b, off, n = loc
        return self.buckets[b]["flat"][off:off + n].view(like.shape)

    def _in_place(self, bucket) -> bool:
        flat = bucket["flat"]
        base = flat.data_ptr()
        for n, p in bucket["params"]:
            if p.grad is None:
                continue
            _, off, _ = self._where[n]
            if p.grad.data_ptr() != base + 4 * off or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                return False
        return True

    def buckets_ready_after(self, finished_group: int) -> List[int]:
        """Indices of the buckets that are complete once the backward has finished exit group ``finished_group`` (-1: stem)."""
        return [i for i, b in enumerate(self.buckets) if b["ready_after"] == finished_group]

    def allreduce_bucket(self, i: int, weight: float, group: Optional[dist.ProcessGroup] = None, async_op: bool = True,
                         trusted: bool = False) -> None:
        """Scale bucket ``i`` by this rank's shard weight and sum it over the ranks.  ``async_op``: the collective runs on the
        backend's own stream behind everything enqueued on the current stream so far; ``wait()`` joins it.  ``trusted``: the
        caller has just written EVERY gradient of the bucket into its views (the backward's progress callback, before
        autograd has installed them as ``p.grad``): reduce the flat buffer as it stands."""
        if not _active(group):
            return
        b = self.buckets[i]
        if trusted:
            buf, scatter = b["flat"], None
        elif self._in_place(b):
            # parameters of the bucket that received no gradient this step contribute zeros (their slices may hold an old step)
            for n, p in b["params"]:
                if p.grad is None:
                    _, off, cnt = self._where[n]
                    b["flat"][off:off + cnt].zero_()
            buf, scatter = b["flat"], None
        else:  # gradients live outside the flat buffer: gather, reduce, scatter back
            with_grad = [(n, p) for n, p in b["params"] if p.grad is not None]
            if not with_grad:
                return
            buf = torch.cat([p.grad.reshape(-1).float() for _, p in with_grad])
            scatter = with_grad
        if weight != 1.0:
            buf.mul_(weight)
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        self._pending.append((work, buf, scatter))

    def wait(self) -> int:
        """Join every outstanding collective (the current stream then sees the reduced gradients); returns their number."""
        n = len(self._pending)
        for work, buf, scatter in self._pending:
            if work is not None:
                work.wait()
            if scatter is not None:
                off = 0
                for _, p in scatter:
                    cnt = p.grad.numel()
                    p.grad.copy_(buf[off:off + cnt].view_as(p.grad))
                    off += cnt
        self._pending = []
        return n

    def adopt_views(self, i: int) -> int:
        """After ``wait()``: make sure every ``p.grad`` of bucket ``i`` holds the REDUCED values of its view.  A bucket reduced
        from the backward's callback was reduced before autograd installed the views; if autograd installed a copy instead (a
        tensor hook, another live reference), that copy is stale: overwrite it.  Returns the number of gradients repaired."""
        b, fixed = self.buckets[i], 0
        base = b["flat"].data_ptr()
        for n, p in b["params"]:
            if p.grad is None:
                continue
            _, off, cnt = self._where[n]
            if p.grad.data_ptr() != base + 4 * off:
                p.grad.copy_(b["flat"][off:off + cnt].view_as(p.grad))
                fixed += 1
        return fixed

    def allreduce_all(self, weight: float, group: Optional[dist.ProcessGroup] = None) -> int:
        """Every bucket, in completion order, then ``wait()``: the form for a backward that did not report its progress."""
        for i in range(len(self.buckets)):
            self.allreduce_bucket(i, weight, group)
        return self.wait()


def allreduce_gradients(params: Iterable[torch.nn.Parameter], b_local: int, bucket_bytes: int = 64 << 20,
                        group: Optional[dist.ProcessGroup] = None) -> int:
    """Stand-alone form for gradients that are ordinary tensors (any module, no flat buckets): after the local backward,
    average ``p.grad`` over the ranks, weighted by the ranks' utterance counts.  Gradients are flattened into buckets of
    ``bucket_bytes`` in reverse parameter order (the order the backward produced them), each bucket ONE all-reduce, all of
    them in flight together.  The shard weight is exchanged on EVERY call (``shard_weight_tensor``: one float, every rank
    always enters the collective, so shard sizes may change from step to step) and stays on the device: no per-step host
    read-back.  Returns the number of gradient collectives.  BatchNorm statistics stay per replica.
    Models that train through ``eec_train_backward`` use ``GradBuckets`` instead (in-place buffers, overlapped)."""
    if not _active(group):
        return 0
    with_grad = [p for p in params if p.grad is not None]
    if not with_grad:
        return 0
    w = shard_weight_tensor(b_local, with_grad[0].grad.device, group)
    pending, bucket, size = [], [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bucket]).mul_(w)
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, bucket))
        bucket, size = [], 0

    for p in reversed(with_grad):
        bucket.append(p)
        size += p.grad.numel() * p.grad.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    for work, flat, plist in pending:
        work.wait()
        off = 0
        for p in plist:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return len(pending)
