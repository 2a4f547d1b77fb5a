"""Adaptive density control on the device: the counterpart of ``GSModel``'s
densification methods (reference gsplat/gsmodel.py:170-183, 214-338).

Same method names, arguments and in-place effects on ``params`` (dict name -> leaf
tensor) and ``optimizer`` (``torch.optim.Adam`` or ``optim.FusedAdam``) as the reference:

    ctl = DensityControl(scene_size, max_steps)
    ctl.update_density_info(us.grad, mask)          # after each backward   (gsmodel.py:214-230)
    ctl.update_gaussian_density(params, optimizer)  # prune / clone / split (gsmodel.py:232-317)
    ctl.reset_alpha(params, optimizer)              #                       (gsmodel.py:319-330)
    ctl.update_pws_lr(optimizer)                    #                       (gsmodel.py:332-338)

What differs from the reference is the execution: one classify launch, one scan launch, a
16-byte read-back (the new row count sizes the allocations) and ONE compaction launch that
moves every parameter row and both Adam moments once (``csrc/egs_density.hip``), instead of
~40 boolean-mask gathers and ``torch.cat``s.  The split offsets are drawn from a counter-based
generator keyed by ``(seed, round, row)``, so data-parallel replicas that hold the same
(all-reduced) statistics produce identical Gaussians without any broadcast.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .optim import NAMES


def _logit(x):
    return math.log(x / (1 - x))


def expon_lr(step, lr_init, lr_final, max_steps, delay_steps=0, delay_mult=1.0):
    """Log-linear learning-rate decay with optional warm-up (restates utils.py:7-44)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    rate = 1.0
    if delay_steps > 0:
        rate = delay_mult + (1 - delay_mult) * math.sin(0.5 * math.pi * min(max(step / delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def _pset(tensors):
    return _lib.EgsGaussianParams(*[t.data_ptr() if t is not None and t.numel() else None for t in tensors])


class DensityControl:
    def __init__(self, scene_size: float, max_steps: int, seed: int = 0):
        # gsmodel.py:170-183
        self.grad_threshold = 4e-7
        self.scale_threshold = 0.01 * scene_size
        self.alpha_threshold = 0.005
        self.big_threshold = 0.1 * scene_size
        self.reset_alpha_val = 0.01
        self.scene_size = scene_size
        self.max_steps = max_steps
        self.iteration = 0
        self.seed = seed
        self.round = 0                 # number of densifications done: RNG stream of the next one
        self.grad_accum = None         # [N] float32
        self.cunt = None               # [N] int32 (the reference's spelling)

    # -- statistics ---------------------------------------------------------------------------
    @torch.no_grad()
    def update_density_info(self, dloss_dus: torch.Tensor, mask: torch.Tensor):
        """Accumulate ||dL/du|| and visibility of one view (gsmodel.py:214-230)."""
        lib = _lib.load()
        n = dloss_dus.shape[0]
        dus = dloss_dus.detach().reshape(n, 2).contiguous().float()
        vis = mask.detach().reshape(n).contiguous()
        if vis.dtype != torch.bool and vis.dtype != torch.uint8:
            vis = vis != 0
        first = self.grad_accum is None
        if first:
            self.grad_accum = torch.empty(n, dtype=torch.float32, device=dus.device)
            self.cunt = torch.empty(n, dtype=torch.int32, device=dus.device)
        elif self.grad_accum.shape[0] != n:
            raise ValueError("density statistics hold %d rows, got %d" % (self.grad_accum.shape[0], n))
        _lib.check(lib.egs_density_accumulate(n, dus.data_ptr(), vis.data_ptr(), int(first),
                                              self.grad_accum.data_ptr(), self.cunt.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))

    def set_density_info(self, grad_accum: torch.Tensor, count: torch.Tensor):
        """Install statistics accumulated elsewhere (e.g. all-reduced over data-parallel ranks)."""
        self.grad_accum = grad_accum.reshape(-1).contiguous().float()
        self.cunt = count.reshape(-1).contiguous().to(torch.int32)

    # -- prune / clone / split ------------------------------------------------------------------
    @torch.no_grad()
    def update_gaussian_density(self, params, optimizer, unit_noise: torch.Tensor = None, verbose: bool = False):
        """gsmodel.py:232-317.  Mutates ``params`` and ``optimizer`` like the reference; returns the
        report the reference prints: dict(pruned, cloned, splited, total)."""
        lib = _lib.load()
        if self.grad_accum is None:
            raise RuntimeError("update_gaussian_density needs update_density_info first")
        groups = {g["name"]: g for g in optimizer.param_groups}
        missing = [k for k in NAMES if k not in params or k not in groups]
        if missing:
            raise ValueError("params/optimizer lack the groups %s" % missing)
        cur = [params[k] for k in NAMES]
        n = cur[0].shape[0]
        dev = cur[0].device
        for k, t, w in zip(NAMES, cur, (3, 3, None, 1, 3, 4)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == n and
                    (w is None or t.numel() == n * w)):
                raise ValueError("parameter %s: expected contiguous float32 device tensor [%d, %s]" % (k, n, w))
        if self.grad_accum.shape[0] != n:
            raise ValueError("density statistics hold %d rows, model has %d" % (self.grad_accum.shape[0], n))
        hw = cur[2].shape[1] if cur[2].dim() == 2 else 0
        states = [optimizer.state.get(groups[k]["params"][0], None) for k in NAMES]
        has_state = [s is not None and "exp_avg" in s for s in states]
        if any(has_state) and not all(has_state):
            raise ValueError("optimizer state exists for some groups only")
        has_state = all(has_state)
        stream = torch.cuda.current_stream().cuda_stream

        ws = torch.empty(lib.egs_densify_ws_bytes(n), dtype=torch.uint8, device=dev)
        cls = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        totals = torch.empty(4, dtype=torch.int32, device=dev)
        _lib.check(lib.egs_densify_plan(n, cur[3].data_ptr(), cur[4].data_ptr(), self.grad_accum.data_ptr(),
                                        self.cunt.data_ptr(), _logit(self.alpha_threshold),
                                        math.log(self.big_threshold), self.grad_threshold, self.scale_threshold,
                                        cls.data_ptr(), ws.data_ptr(), ws.numel(), totals.data_ptr(), stream))
        n_keep, n_clone, n_split, n_prune = (int(x) for x in totals.tolist())      # the one read-back
        n_out = n_keep + n_clone + n_split

        new = [torch.empty((n_out,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev) for t in cur]
        if has_state:
            old_m = [s["exp_avg"].contiguous() for s in states]
            old_v = [s["exp_avg_sq"].contiguous() for s in states]
            new_m = [torch.empty_like(t) for t in new]
            new_v = [torch.empty_like(t) for t in new]
            sets = [_pset(old_m), _pset(old_v), _pset(new), _pset(new_m), _pset(new_v)]
            ptrs = [C.byref(x) for x in sets]
        else:
            sets = [_pset(new)]
            ptrs = [None, None, C.byref(sets[0]), None, None]
        noise_ptr = None
        if unit_noise is not None:
            unit_noise = unit_noise.to(dev, torch.float32).contiguous()
            if unit_noise.numel() != 3 * n:
                raise ValueError("unit_noise must be [N, 3]")
            noise_ptr = unit_noise.data_ptr()
        src = _pset(cur)
        _lib.check(lib.egs_densify_apply(n, n_keep, n_clone, n_split, hw, cls.data_ptr(), ws.data_ptr(),
                                         C.byref(src), ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], noise_ptr,
                                         self.seed, self.round, stream))
        # hand the new tensors to the optimizer the way prune_params/update_params do (gsmodel.py:132-166)
        for i, k in enumerate(NAMES):
            grp = groups[k]
            old_p = grp["params"][0]
            st = optimizer.state.pop(old_p, None)
            p = torch.nn.Parameter(new[i].requires_grad_(True))
            grp["params"][0] = p
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = new_m[i], new_v[i]
                optimizer.state[p] = st
            params[k] = p
        self.grad_accum = None
        self.cunt = None
        self.round += 1
        report = {"pruned": n_prune, "cloned": n_clone, "splited": n_split, "total": n_out}
        if verbose:
            print("gaussian density update report: pruned %(pruned)d cloned %(cloned)d splited %(splited)d "
                  "total %(total)d" % report)
        return report

    @torch.no_grad()
    def reset_alpha(self, params, optimizer):
        """gsmodel.py:319-330: alphas_raw = min(alphas_raw, logit(0.01)); zero its Adam moments."""
        lib = _lib.load()
        a = params["alphas_raw"]
        grp = [g for g in optimizer.param_groups if g["name"] == "alphas_raw"][0]
        st = optimizer.state.get(grp["params"][0], None)
        m = v = None
        if st is not None and "exp_avg" in st:
            m, v = st["exp_avg"], st["exp_avg_sq"]
        _lib.check(lib.egs_reset_alpha(a.numel(), _logit(self.reset_alpha_val), a.data_ptr(),
                                       m.data_ptr() if m is not None else None,
                                       v.data_ptr() if v is not None else None,
                                       torch.cuda.current_stream().cuda_stream))
        if a.is_cuda:      # nothing saturates any more: the next renders walk their whole lists (fused.expect_long_walks)
            from . import fused as _fused
            _fused.expect_long_walks(a.device, renders=4)

    def update_pws_lr(self, optimizer):
        """gsmodel.py:332-338 with the schedule of gsmodel.py:180-183."""
        lr = expon_lr(self.iteration, 1e-4 * self.scene_size, 1e-6 * self.scene_size, self.max_steps,
                      delay_mult=0.01)
        for g in optimizer.param_groups:
            if g["name"] == "pws":
                g["lr"] = lr
        self.iteration += 1
        return lr
