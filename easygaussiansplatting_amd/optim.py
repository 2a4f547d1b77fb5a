"""Adam for the six Gaussian parameter groups as ONE HIP launch per step.

The reference trains with ``torch.optim.Adam(adam_params, lr=0.000, eps=1e-15)``
(train.py:32) over the groups of gsmodel.py:114-127.  ``FusedAdam`` keeps torch's
public layout -- ``param_groups`` (dicts with ``params``, ``lr``, ``name``) and
``state[param] = {"step", "exp_avg", "exp_avg_sq"}`` -- so the optimizer surgery of
``density.py`` (and any code written against the reference's optimizer) treats both
interchangeably; only ``step()`` differs: all groups are updated by ``egs_adam_step``
(7 x 4 B of HBM traffic per parameter, no temporaries).
"""
from __future__ import annotations

import torch

from . import _lib

NAMES = ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")
REFERENCE_LRS = (0.001, 0.001, 0.001 / 20, 0.05, 0.005, 0.001)          # gsmodel.py:114-127


def adam_groups(params):
    """The reference's per-group learning rates (gsmodel.py:114-127)."""
    return [{"params": [params[k]], "lr": lr, "name": k} for k, lr in zip(NAMES, REFERENCE_LRS)]


class FusedAdam:
    """torch.optim.Adam semantics (no weight decay, no amsgrad) on libegs_hip.so."""

    def __init__(self, param_groups, lr=0.0, betas=(0.9, 0.999), eps=1e-15):
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g["params"] = list(g["params"])
            self.param_groups.append(g)
        if len(self.param_groups) > 8:
            raise ValueError("FusedAdam handles at most 8 parameter groups per launch")
        self.betas = betas
        self.eps = eps
        self.state = {}

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self, factored_sh=None):
        """``factored_sh`` = ``(rows, scale, pws, low_shs, high_shs | None)``: the step's SH gradient in its factored
        form (``dist_views.FactoredShGrad.take()``: dL/dcolour [N,3] and camera centre per view; ``scale`` = 1 / ranks
        for a mean over ranks, 1 for a sum) -- the SH tensors (``.grad`` None) are then updated by
        ``egs_adam_sh_factored``, which forms each Gaussian's gradient row in LDS instead of reading it from memory."""
        lib = _lib.load()
        # 1. validate EVERYTHING first: nothing of the optimizer's state may move for a step that is then refused (a
        #    bumped ``step`` without an update would leave wrong bias corrections behind: ADVICE r4)
        plain, keep = [], []      # keep: contiguous gradient copies, alive until the launch is enqueued
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise ValueError("FusedAdam needs contiguous float32 device parameters")
                grad = p.grad.contiguous()
                keep.append(grad)
                plain.append((p, grad, float(g["lr"])))
        sh = []
        if factored_sh is not None:
            rows, scale, pws, low, high = factored_sh
            for t in ((low, high) if (high is not None and high.shape[1] > 0) else (low,)):
                if t.grad is not None:
                    raise ValueError("FusedAdam.step(factored_sh=...): an SH tensor already holds a .grad")
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise ValueError("FusedAdam needs contiguous float32 device parameters")
                lrs = [float(g["lr"]) for g in self.param_groups if any(q is t for q in g["params"])]
                if len(lrs) != 1:
                    raise ValueError("FusedAdam.step(factored_sh=...): an SH tensor that is not in exactly one group")
                sh.append((t, lrs[0]))
        # 2. state and step counters; a launch that is refused after all (the C side checks alignment / counts) rolls back
        # the counters of ITS groups and of those not launched yet -- tensors an earlier launch already updated keep
        # their new step (a retry must not apply the same step to them twice)
        bumped = []

        def state_of(t):
            st = self.state.get(t)
            if st is None:
                st = self.state[t] = {"step": 0, "exp_avg": torch.zeros_like(t), "exp_avg_sq": torch.zeros_like(t)}
            st["step"] = int(st["step"]) + 1
            bumped.append(st)
            return st
        stream = torch.cuda.current_stream().cuda_stream
        try:
            recs = []
            for p, grad, lr in plain:
                st = state_of(p)
                recs.append(_lib.EgsAdamGroup(p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(),
                                              st["exp_avg_sq"].data_ptr(), p.numel(), lr, st["step"]))
            if factored_sh is not None:
                n = pws.shape[0]
                groups = []
                for t, lr in sh:
                    st = state_of(t)
                    groups.append(_lib.EgsAdamGroup(t.data_ptr(), None, st["exp_avg"].data_ptr(),
                                                    st["exp_avg_sq"].data_ptr(), t.numel(), lr, st["step"]))
                K = low.shape[1] + (high.shape[1] if high is not None else 0)
                _lib.check(lib.egs_adam_sh_factored(
                    n, K, rows.shape[0], pws.data_ptr(), rows.data_ptr(), rows.shape[1], float(scale), groups[0],
                    groups[1] if len(groups) > 1 else None, self.betas[0], self.betas[1], self.eps, stream))
                del bumped[len(plain):]          # enqueued: the SH tensors' counters stand
            for i in range(0, len(recs), 8):
                chunk = recs[i:i + 8]
                arr = (_lib.EgsAdamGroup * len(chunk))(*chunk)
                _lib.check(lib.egs_adam_step(len(chunk), arr, self.betas[0], self.betas[1], self.eps, stream))
                for j in range(i, i + len(chunk)):
                    bumped[j] = None             # enqueued
        except BaseException:
            for st in bumped:
                if st is not None:
                    st["step"] -= 1
            raise
