"""One camera view per GPU, gradient exchange over RCCL/xGMI.

The reference is single-process, single-GPU (train.py:48-57 renders one view
per optimizer step).  The only place the rasterizer path shards is BY VIEW:
every rank keeps a full replica of the Gaussian parameters, renders its own
view forward+backward, and the step ends with ONE exchange:

* all-reduce(mean) of the parameter gradients -- pws 3 + shs 48 + alphas 1 +
  scales 3 + rots 4 = 59 fp32 per Gaussian (236 MB at N = 1 M);
* all-reduce(sum) of the per-view densification statistics the reference
  accumulates in ``GSModel.update_density_info`` (gsmodel.py:214-230): the norm
  of dL/du per Gaussian (the norm is per view, so norms are reduced, not dus)
  and the visibility count.

No data-path collective exists anywhere else (binning/sort/draw are per view).
``torch.distributed`` backend "nccl" is RCCL on ROCm; the CPU tests run the same
code over "gloo".
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

PARAM_ORDER = ("pws", "shs", "alphas", "scales", "rots")
GRAD_FLOATS_PER_GAUSSIAN = {"pws": 3, "shs": 48, "alphas": 1, "scales": 3, "rots": 4}


def views_for_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Static round-robin assignment of camera views to ranks (view v -> rank v % world)."""
    return [v for v in range(n_views) if v % world == rank]


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_(tensors: Sequence[torch.Tensor], group=None) -> None:
    """In-place mean over ranks of every tensor, issued as asynchronous
    collectives and waited together (the 192-MB SH gradient dominates)."""
    world = _world(group)
    if world == 1:
        return
    backend = dist.get_backend(group)
    avg = backend == "nccl"  # RCCL implements ncclAvg; gloo has no AVG
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    handles = [dist.all_reduce(t, op=op, group=group, async_op=True) for t in tensors]
    for h in handles:
        h.wait()
    if not avg:
        for t in tensors:
            t.div_(world)


def allreduce_sum_(tensors: Sequence[torch.Tensor], group=None) -> None:
    if _world(group) == 1:
        return
    handles = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in tensors]
    for h in handles:
        h.wait()


def exchange_gradients(params: Dict[str, torch.Tensor], group=None) -> None:
    """Mean-reduce ``p.grad`` of the five parameter groups across ranks."""
    grads = []
    for k in PARAM_ORDER:
        g = params[k].grad
        if g is None:
            raise RuntimeError("parameter %r has no gradient to exchange" % k)
        grads.append(g)
    allreduce_mean_(grads, group)


def density_stats(dloss_dus: torch.Tensor, mask: torch.Tensor, group=None):
    """Per-view ||dL/du|| and visibility, summed over ranks (gsmodel.py:219-228)."""
    grad_norm = torch.norm(dloss_dus.reshape(-1, 2), dim=-1)
    grad_norm = torch.where(mask, grad_norm, torch.zeros_like(grad_norm))
    count = mask.to(torch.int32)
    allreduce_sum_([grad_norm, count], group)
    return grad_norm, count


def grad_exchange_bytes(n_gaussians: int) -> int:
    return 4 * n_gaussians * sum(GRAD_FLOATS_PER_GAUSSIAN.values())
