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


def flat_grad_buffer(tensors):
    """One 1-D tensor over the storage behind the ``.grad`` of the given parameter tensors, when they all
    came out of one ``backward`` call of this module and tile that storage exactly (up to the 16-B
    padding between slices); else None."""
    grads = [t.grad for t in tensors]
    if any(g is None or not g.is_contiguous() or g.dtype != torch.float32 for g in grads):
        return None
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads):
        return None
    total = st.nbytes() // 4
    at = 0
    for off, cnt in sorted((g.storage_offset(), g.numel()) for g in grads):
        if not (at <= off < at + 4):      # slices may be padded to 16 B: gaps of up to 3 floats
            return None
        at = off + cnt
    if not (at <= total < at + 4):
        return None
    return torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, 0, (total,))


def coalesce_grads(tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """``[flat]`` when the gradients of ``tensors`` are slices of one buffer (the fused backward allocates
    them that way: one 236-MB collective instead of five or six latency-bound ones), else the gradients."""
    flat = flat_grad_buffer(tensors)
    return [flat] if flat is not None else [t.grad for t in tensors]


def exchange_gradients(params: Dict[str, torch.Tensor], group=None, names: Optional[Iterable[str]] = None) -> None:
    """Mean-reduce ``p.grad`` of the parameter groups (default: the five activated ones) across ranks."""
    names = tuple(names) if names is not None else PARAM_ORDER
    for k in names:
        if params[k].grad is None:
            raise RuntimeError("parameter %r has no gradient to exchange" % k)
    if _world(group) == 1:
        return
    allreduce_mean_(coalesce_grads([params[k] for k in names]), group)


def density_stats(dloss_dus: torch.Tensor, mask: torch.Tensor, group=None):
    """Per-view ||dL/du|| and visibility, summed over ranks (gsmodel.py:219-228)."""
    grad_norm = torch.norm(dloss_dus.reshape(-1, 2), dim=-1)
    grad_norm = torch.where(mask, grad_norm, torch.zeros_like(grad_norm))
    count = mask.to(torch.int32)
    allreduce_sum_([grad_norm, count], group)
    return grad_norm, count


def grad_exchange_bytes(n_gaussians: int) -> int:
    return 4 * n_gaussians * sum(GRAD_FLOATS_PER_GAUSSIAN.values())
