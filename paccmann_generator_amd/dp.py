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

    The buffer covers the FIXED list ``params`` on every rank: a parameter without a gradient on
    this rank takes part as zeros (and receives the sum), so ranks whose backward reached different
    sets of parameters -- or did not run at all -- still exchange buffers of one size.  Returns the
    number of floats reduced."""
    ps = list(params)
    reduce_step_(ps, (), group=group, keep_none=False)
    return sum(p.numel() for p in ps)


def reduce_step_(params: Sequence[torch.nn.Parameter], scalars: Sequence[float] = (), group=None,
                 keep_none: bool = True) -> List[float]:
    """The ONE data-path collective of a data-parallel REINFORCE step.

    Flat fp32 buffer = [grad of every parameter in ``params`` (zeros where ``p.grad is None``) |
    one has-gradient flag per parameter | ``scalars``], all-reduced (sum) in a single call.  The
    summed gradients are written back to ``p.grad``; with ``keep_none`` a parameter that had no
    gradient on ANY rank keeps ``grad = None`` (what a single process would have left).  Every rank
    must call this with the same parameter list, including ranks with nothing to contribute (empty
    shard, no valid molecule): they pass gradients of None and zero scalars.  Returns the summed
    scalars."""
    params = list(params)
    if not params and not scalars:
        return []
    ref = next((p for p in params), None)
    device = ref.device if ref is not None else torch.device('cpu')
    pieces = [(p.grad.detach().reshape(-1).to(torch.float32) if p.grad is not None
               else torch.zeros(p.numel(), dtype=torch.float32, device=device)) for p in params]
    tail = torch.tensor([0.0 if p.grad is None else 1.0 for p in params] + [float(x) for x in scalars],
                        dtype=torch.float32, device=device)
    flat = torch.cat(pieces + [tail])
    all_reduce_(flat, group=group)
    n_tail = tail.numel()
    tail_h = flat[flat.numel() - n_tail:].cpu().tolist()
    flags, out = tail_h[:len(params)], tail_h[len(params):]
    off = 0
    for p, has in zip(params, flags):
        n = p.numel()
        if has > 0.0 or not keep_none:
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
        off += n
    return out


def data_parallel_step(dec_params: Sequence[torch.nn.Parameter], enc_params: Sequence[torch.nn.Parameter],
                       optimizer, backward_fn: Callable[[], float], num_mols: int, reward_sum: float,
                       grad_clipping: Optional[float], group=None) -> Tuple[float, float, int]:
    """One data-parallel REINFORCE update on this rank's shard (reinforce_omics.py:289-341 with the
    mean over valid molecules taken over ALL ranks).  Device-agnostic: the HIP operator (or, in the
    CPU tests, the oracle) comes in through ``backward_fn``.

    ``backward_fn()`` runs forward + backward with scale 1 on the ``num_mols`` valid molecules of this
    rank, leaves gradient SUMS in ``.grad`` and returns the loss sum; it is not called when
    ``num_mols == 0``.  Then ONE all-reduce carries [gradients | flags | loss sum | reward sum | count]
    (reduce_step_); every rank takes part, also one with nothing to contribute.  Everything is
    divided by the global count, clip and the optimizer step run replicated.  When no rank has a valid
    molecule the update is skipped everywhere.

    Encoder gradients: the optimizer holds decoder parameters only (reinforce.py:44-45), so the
    reference never zeroes them and they carry over from step to step.  Only THIS step's
    contribution is summed over the ranks; the carried part is set aside before the backward pass
    and added back afterwards (summing it too would multiply it by the world size every step).

    Returns (mean reward, loss, global valid count)."""
    dec_params, enc_params = list(dec_params), list(enc_params)
    optimizer.zero_grad()
    stash = [p.grad for p in enc_params]
    for p in enc_params:
        p.grad = None
    loss_sum = float(backward_fn()) if num_mols > 0 else 0.0
    loss_sum, reward_sum, n_glob = reduce_step_(dec_params + enc_params, (loss_sum, float(reward_sum), float(num_mols)),
                                                group=group)
    n_glob = int(round(n_glob))
    if n_glob == 0:
        for p, g in zip(enc_params, stash):
            p.grad = g
        optimizer.zero_grad()
        return 0.0, 0.0, 0
    inv = 1.0 / n_glob
    fresh = [p.grad for p in dec_params + enc_params if p.grad is not None]
    if fresh:
        torch._foreach_mul_(fresh, inv)
    for p, g in zip(enc_params, stash):
        if g is not None:
            p.grad = g if p.grad is None else p.grad.add_(g)
    if grad_clipping is not None:
        torch.nn.utils.clip_grad_norm_(dec_params + enc_params, grad_clipping)
    optimizer.step()
    return reward_sum * inv, loss_sum * inv, n_glob
