"""Data-parallel sharding of the latent batch: one process per GPU, RCCL over xGMI.

The reference has no distributed code; BASELINE.json's north_star shards the batch of
latent samples over the GPUs of a node and all-reduces the policy gradients.  Sequences
are independent, so the only exchange per REINFORCE step is

  * the count of valid molecules (the reference's ``mean`` over valid molecules,
    reinforce_omics.py:330-332, becomes ``sum / B_valid_global``), and
  * ONE all-reduce (sum) of the flat gradient buffer (5.0 MB at configs[1]).

Everything here is device-agnostic torch.distributed, so the same code runs over RCCL
("nccl" backend on ROCm) on the GPUs and over gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def _stage_on_host(t: torch.Tensor, group=None) -> bool:
    """gloo moves device tensors through the host itself, and that path hangs on this ROCm stack:
    stage explicitly.  RCCL ("nccl") reduces device memory in place."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def all_reduce_(t: torch.Tensor, op=None, group=None) -> torch.Tensor:
    """In-place all-reduce that works for device tensors under both RCCL and gloo."""
    op = dist.ReduceOp.SUM if op is None else op
    if _stage_on_host(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n`` sequences owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_sum(value: float, device, group=None) -> float:
    """Sum of a host scalar over the ranks (used for B_valid_global)."""
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    all_reduce_(t, group=group)
    return float(t.item())


def all_reduce_gradients_(params: Iterable[torch.nn.Parameter], group=None) -> int:
    """Sum ``p.grad`` over the ranks with one collective on one flat fp32 buffer.

    Parameters without a gradient on this rank take part as zeros when any rank could
    have one is NOT supported: every rank must hold the same set of gradients (true for
    the REINFORCE step, where every rank runs the same operator).  Returns the number of
    floats reduced."""
    ps: List[torch.nn.Parameter] = [p for p in params if p.grad is not None]
    if not ps:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    all_reduce_(flat, group=group)
    off = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return off
