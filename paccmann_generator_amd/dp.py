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

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

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


def broadcast_(t: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """In-place broadcast that works for device tensors under both RCCL and gloo."""
    if _stage_on_host(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
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


def global_max(value: int, device, group=None) -> int:
    """Largest of a host integer over the ranks (the global ``T_max`` of a sharded batch: the
    reference pads every molecule to the longest VALID one of the whole batch and does not mask the
    padded positions, reinforce_omics.py:304,322-332, so every rank has to run that many steps)."""
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    all_reduce_(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def all_reduce_gradients_(params: Iterable[torch.nn.Parameter], group=None) -> int:
    """Sum ``p.grad`` over the ranks with one collective on one flat fp32 buffer.

    The buffer covers the FIXED list ``params`` on every rank: a parameter without a gradient on
    this rank takes part as zeros, so ranks whose backward reached different sets of parameters --
    or did not run at all -- still exchange buffers of one size.  A parameter that has a gradient
    on NO rank keeps ``grad = None`` (the has-gradient flags travel in the same buffer), so that an
    optimizer with weight decay leaves it alone exactly as the single-process run does.  Returns
    the number of floats reduced."""
    ps = list(params)
    local = [p.grad is not None for p in ps]
    flat, n_tail = _reduce_flat(ps, [], group)
    if all(local):
        take = local                       # no host synchronisation needed
    else:
        flags = flat[flat.numel() - n_tail:].cpu().tolist()
        take = [f > 0.0 for f in flags[:len(ps)]]
    _scatter_grads(ps, flat, take)
    return sum(p.numel() for p in ps)


def _reduce_flat(params: Sequence[torch.nn.Parameter], scalars: Sequence, group):
    """[grad of every parameter (zeros where None) | has-gradient flag per parameter | scalars],
    all-reduced (sum) in ONE call.  ``scalars`` may be host numbers or 0-dim device tensors (no
    host synchronisation either way).  Returns (flat, number of tail entries)."""
    ref = params[0] if params else None
    device = ref.device if ref is not None else torch.device('cpu')
    pieces = [(p.grad.detach().reshape(-1).to(torch.float32) if p.grad is not None
               else torch.zeros(p.numel(), dtype=torch.float32, device=device)) for p in params]
    host = [0.0 if p.grad is None else 1.0 for p in params] + [0.0 if torch.is_tensor(x) else float(x) for x in scalars]
    tail = torch.tensor(host, dtype=torch.float32).to(device, non_blocking=True)
    for i, x in enumerate(scalars):
        if torch.is_tensor(x):
            tail[len(params) + i] = x.detach().to(device=device, dtype=torch.float32).reshape(())
    flat = torch.cat(pieces + [tail])
    all_reduce_(flat, group=group)
    return flat, tail.numel()


def _scatter_grads(params, flat, take) -> None:
    """``p.grad`` becomes a view of its slice of ``flat`` (no copies) for every parameter with take[i]."""
    off = 0
    for p, t in zip(params, take):
        n = p.numel()
        if t:
            g = flat[off:off + n].view_as(p)
            p.grad = g if p.dtype == torch.float32 else g.to(p.dtype)
        off += n


def data_parallel_step(dec_params: Sequence[torch.nn.Parameter], enc_params: Sequence[torch.nn.Parameter],
                       optimizer, backward_fn: Callable[[], object], num_mols: int, reward_sum,
                       grad_clipping: Optional[float], group=None):
    """One data-parallel REINFORCE update on this rank's shard (reinforce_omics.py:289-341 with the
    mean over valid molecules taken over ALL ranks).  Device-agnostic: the HIP operator (or, in the
    CPU tests, the oracle) comes in through ``backward_fn``.

    ``backward_fn()`` runs forward + backward with scale 1 on the ``num_mols`` valid molecules of this
    rank, leaves gradient SUMS in ``.grad`` and returns the loss sum; it is not called when
    ``num_mols == 0``.  The loss sum and ``reward_sum`` may be numbers or 0-dim tensors
    (``reward_sum`` also a callable evaluated after ``backward_fn``).  Then ONE all-reduce carries [gradients | has-gradient flags |
    loss sum | reward sum | count]; every rank takes part, also one with nothing to contribute.
    Everything is divided by the global count, clip and the optimizer step run replicated.  When no
    rank has a valid molecule the update is skipped everywhere.

    A rank that has valid molecules needs no host synchronisation here: its own count tells it that
    the global count is positive, and a parameter without a gradient after its backward pass has
    none on any rank (same graph everywhere).  Only a rank with nothing to contribute reads the
    reduced flags and count back.

    Encoder gradients: the optimizer holds decoder parameters only (reinforce.py:44-45), so the
    reference never zeroes them and they carry over from step to step.  Only THIS step's
    contribution is summed over the ranks; the carried part is set aside before the backward pass
    and added back afterwards (summing it too would multiply it by the world size every step).

    Returns (mean reward, loss) as 0-dim tensors on the parameters' device -- (0, 0) when the update
    was skipped -- and the global valid count (a 0-dim tensor, or the int 0)."""
    dec_params, enc_params = list(dec_params), list(enc_params)
    params = dec_params + enc_params
    optimizer.zero_grad()
    stash = [p.grad for p in enc_params]
    for p in enc_params:
        p.grad = None
    loss_sum = backward_fn() if num_mols > 0 else 0.0
    if callable(reward_sum):            # known only after the backward pass (pipelined host scoring)
        reward_sum = reward_sum()
    flat, n_tail = _reduce_flat(params, (loss_sum, reward_sum, float(num_mols)), group)
    tail = flat[flat.numel() - n_tail:]
    n_glob = tail[-1].clone()
    if num_mols > 0:
        take = [p.grad is not None for p in params]
    else:
        host = tail.cpu().tolist()
        if int(round(host[-1])) == 0:
            for p, g in zip(enc_params, stash):
                p.grad = g
            optimizer.zero_grad()
            zero = torch.zeros((), dtype=torch.float32, device=flat.device)
            return zero, zero.clone(), 0
        take = [f > 0.0 for f in host[:len(params)]]
    inv = 1.0 / n_glob
    flat.mul_(inv)                      # gradients, loss sum and reward sum -> means (flags/count unused after)
    _scatter_grads(params, flat, take)
    for p, g in zip(enc_params, stash):
        if g is not None:
            p.grad = g if p.grad is None else g.add_(p.grad)
    if grad_clipping is not None:
        torch.nn.utils.clip_grad_norm_(params, grad_clipping)
    optimizer.step()
    return tail[-2].clone(), tail[-3].clone(), n_glob
