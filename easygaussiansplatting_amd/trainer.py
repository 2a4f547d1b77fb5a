"""Data-parallel training step: the counterpart of the reference's ``train.py:30-83``
on the MI355X path (SURVEY.md §8a', §8e).

What it keeps from the reference
* the raw parameterisation and activations (gsplat/utils.py:121-150: sigmoid alphas,
  exp scales, normalised quaternions, SH = cat(low, high); gsmodel.py:96-129);
* Adam with the reference's per-group learning rates (gsmodel.py:114-127) and
  eps = 1e-15 (train.py:32), the exponential position-lr schedule (utils.py:7-44);
* the loss 0.8 L1 + 0.2 (1 - SSIM) (pytorch_ssim.py:63-66) -- through the fused HIP loss;
* checkpoints as a structured ``.npy`` of ACTIVATED parameters with the reference's
  record dtype (gau_io.py:7-12, 141-156).

What is new (the reference renders one view per optimizer step on one GPU)
* a step renders ``views_per_step`` views: each rank takes its share
  (dist_views.views_for_rank), accumulates the mean gradient over its local views,
  then ONE RCCL all-reduce (mean) of the 59 gradient floats per Gaussian and one
  all-reduce (sum) of the densification statistics;
* the optimizer is ``optim.FusedAdam`` (one HIP launch for all six groups) unless
  ``fused_adam=False``; densification / alpha reset (gsmodel.py:232-330) run on the device
  through ``density.DensityControl`` and are replica-consistent: every rank holds the same
  all-reduced statistics and the split offsets are a pure function of (seed, round, row).
"""
from __future__ import annotations

import contextlib
import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import dist_views as DV
from . import fused as _fused
from .density import DensityControl, expon_lr
from .function import Camera, GSFunction, GSRawFunction, RenderOptions
from .loss import gau_loss, gau_loss_with_grad
from .optim import FusedAdam, adam_groups
from .scene import gsdata_type

SH_C0 = 0.28209479177387814


def raw_params_from_scene(scene, device="cuda", clamp_alpha: bool = True) -> Dict[str, torch.Tensor]:
    """gsmodel.py:96-113: un-activated leaf tensors; SH split into degree 0 (low) and the rest (high).
    ``clamp_alpha`` keeps the logit finite for opacities of exactly 0 or 1 (the reference does not clamp)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
    shs = t(scene.shs)
    n = shs.shape[0]
    high = torch.full((n, 45), 0.001, device=device)
    if shs.shape[1] > 3:
        high[:, : shs.shape[1] - 3] = shs[:, 3:]
    alphas = t(scene.alphas).reshape(-1, 1)
    if clamp_alpha:
        alphas = alphas.clamp(1e-4, 1 - 1e-4)
    p = {"pws": t(scene.pws), "low_shs": shs[:, :3].contiguous(), "high_shs": high,
         "alphas_raw": torch.log(alphas / (1 - alphas)), "scales_raw": torch.log(t(scene.scales)),
         "rots_raw": t(scene.rots)}
    for v in p.values():
        v.requires_grad_(True)
    return p


def activate(p):
    """utils.py:129-150 / gsmodel.py:198-207."""
    return (p["pws"], torch.cat((p["low_shs"], p["high_shs"]), dim=1), torch.sigmoid(p["alphas_raw"]),
            torch.exp(p["scales_raw"]), torch.nn.functional.normalize(p["rots_raw"]))


def make_optimizer(p, fused=True):
    """train.py:32 over the groups of gsmodel.py:114-127."""
    cls = FusedAdam if fused else torch.optim.Adam
    return cls(adam_groups(p), lr=0.0, eps=1e-15)


class Trainer:
    def __init__(self, scene, cameras: Sequence, gt_images: Sequence[torch.Tensor], max_steps: int,
                 scene_size: float = 1.0, device="cuda", fused_adam: bool = True, seed: int = 0,
                 fused_activations: bool = True, view_streams: int = 4, factored_sh: bool = True, mode: str = "fused"):
        self.device = device
        # how THIS trainer's renders are evaluated (function.RenderOptions.mode; "ops" needs fused_activations=False):
        # carried by every call, never by a process-wide switch -- two trainers in one process may differ
        self.mode = mode
        if mode != "fused" and fused_activations:
            raise ValueError("Trainer(mode=%r) needs fused_activations=False (GSRawFunction is the fused path)" % (mode,))
        # a rank's views of a step go round-robin to this many HIP streams (dist_views.ViewStreams); 1 = one after
        # the other on the caller's stream
        self.view_streams = max(1, int(view_streams))
        self._vs = None
        self.factored_sh = bool(factored_sh)     # see step(): the SH gradient of a step kept as dL/dcolour per view
        self._fx = None
        self._us = {}
        # True: GSRawFunction (activations inside the HIP kernels); False: torch activations + GSFunction,
        # the reference's structure (gsmodel.py:198-210)
        self.fused_activations = fused_activations
        self.params = raw_params_from_scene(scene, device)
        self.opt = make_optimizer(self.params, fused_adam)
        self.density = DensityControl(scene_size, max_steps, seed)
        self.cams = [c if isinstance(c, Camera) else Camera.from_scene(c, device) for c in cameras]
        self.gts = list(gt_images)
        self.max_steps = max_steps
        self.scene_size = scene_size
        self.iteration = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        n = self.params["pws"].shape[0]
        self.grad_accum = torch.zeros(n, device=device)             # gsmodel.py:214-230 statistics
        self.vis_count = torch.zeros(n, dtype=torch.int32, device=device)
        self.redone_steps = 0        # steps rendered twice because a view outgrew the enqueue-ahead buffers

    _KEYS = ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")

    def _us_leaf(self, lane, n):
        """The zero ``us`` leaf of gsmodel.py:198-199, one per lane, kept across steps (only its ``.grad`` matters:
        dL/du of the view just rendered); a fresh ``torch.zeros`` per view is a fill kernel per view."""
        u = self._us.get(lane)
        if u is None or u.shape[0] != n:
            u = self._us[lane] = torch.zeros((n, 2), device=self.device, requires_grad=True)
        u.grad = None
        return u

    def _render_views(self, mine, n_views, opts=None):
        """forward + loss + backward of this rank's views; leaves accumulate the mean over ALL views of the step.
        Two or more views: dealt to ``view_streams`` HIP streams (one gradient accumulator and one set of
        statistics per stream, added up at the end)."""
        n = self.params["pws"].shape[0]
        lanes = min(self.view_streams, len(mine)) if str(self.device).startswith("cuda") else 1
        vs = None
        if lanes > 1:
            if self._vs is None or self._vs.n != lanes or \
                    any(a is not self.params[k] for a, k in zip(self._vs.params, self._KEYS)):
                self._vs = DV.ViewStreams([self.params[k] for k in self._KEYS], lanes)
            vs = self._vs
            vs.begin()
        loss_sum = [torch.zeros((), device=self.device) for _ in range(lanes)]
        gnorm = [torch.zeros(n, device=self.device) for _ in range(lanes)]
        count = [torch.zeros(n, dtype=torch.int32, device=self.device) for _ in range(lanes)]
        if vs is not None:      # the accumulators above were zeroed on the caller's stream: the lanes start after that
            for s in vs.streams:
                s.wait_stream(torch.cuda.current_stream())
        for i, v in enumerate(mine):
            k = i % lanes
            with (vs.lane(i) if vs is not None else contextlib.nullcontext(None)) as lv:
                p = dict(zip(self._KEYS, lv)) if lv is not None else self.params
                us = self._us_leaf(k, n)                                             # gsmodel.py:198-199
                if self.fused_activations:
                    image, mask = GSRawFunction.apply(p["pws"], p["low_shs"], p["high_shs"], p["alphas_raw"],
                                                      p["scales_raw"], p["rots_raw"], us, self.cams[v], opts)
                else:
                    image, mask = GSFunction.apply(*activate(p), us, self.cams[v], opts)
                # loss / n_views (train.py:52-57 with the mean over the step's views): the loss kernels produce the
                # scaled dL/dimage themselves, backward starts at the image
                stats, dimage = gau_loss_with_grad(image.detach(), self.gts[v], grad_scale=1.0 / n_views)
                image.backward(dimage)
                loss_sum[k] += stats[0]
                with torch.no_grad():                       # per-view ||dL/du|| (undo the 1/len scaling)
                    g = torch.norm(us.grad * n_views, dim=-1)
                    gnorm[k] += torch.where(mask, g, torch.zeros_like(g))
                    count[k] += mask.to(torch.int32)
        if vs is not None:
            vs.finish()                                    # the caller's stream now follows every lane
            for k in range(1, lanes):
                for t in (loss_sum[k], gnorm[k], count[k]):
                    t.record_stream(torch.cuda.current_stream())
                loss_sum[0] += loss_sum[k]; gnorm[0] += gnorm[k]; count[0] += count[k]
        return loss_sum[0], gnorm[0], count[0]

    def step(self, view_ids: Sequence[int], sync: bool = True):
        """One optimizer step on the mean gradient over ``view_ids`` (all ranks pass the same list, which must
        give every rank at least one view: a rank without a view would have no gradient to contribute and
        its peers would wait in the all-reduce for ever).  Returns the mean loss: a float (``sync=True``: the
        ``loss.item()`` of train.py:58, one host-device synchronisation per step) or, with ``sync=False``, a
        0-dim device tensor -- the host then never waits for the GPU inside the step (``fit`` uses this and
        reads the losses once per epoch)."""
        if len(view_ids) < self.world:
            raise ValueError("step() got %d view(s) for %d ranks: every rank needs at least one view per step"
                             % (len(view_ids), self.world))
        mine = [view_ids[i] for i in DV.views_for_rank(len(view_ids), self.rank, self.world)]
        self.opt.zero_grad(set_to_none=True)
        # The views are rendered with deferred validation: the host does not wait for a patch count inside the
        # step.  commit() (one wait for the binning stage of the last view, while its draw and backward
        # kernels are still queued) tells whether some view outgrew the buffers sized from earlier renders;
        # that happens while the trainer still meets new views, and the step is then redone exactly.
        # (from the second local view on the chain-rule kernel adds to the leaves' .grad itself: accumulate_in_kernel)
        # The SH gradient of the step stays factored when that moves fewer bytes (dist_views.FactoredShGrad: a view
        # leaves dL/dcolour [N,3]; 12 B per Gaussian and view are all-gathered instead of 192 B per Gaussian
        # all-reduced, and on one rank V views write 12 V + 192 B instead of accumulating 192-B rows V times)
        fx = None
        vmax = -(-len(view_ids) // self.world)      # rows per rank: the same on every rank (one all-gather)
        # (on one rank it pays from the second view on -- or from the first when the optimizer consumes the factored
        # form directly, FusedAdam: the 192-B rows are then never written at all)
        if self.factored_sh and self.fused_activations and \
                (DV.factored_exchange_pays(self.world, vmax, 3 + self.params["high_shs"].shape[1]) or
                 (self.world == 1 and isinstance(self.opt, FusedAdam))):
            if self._fx is None or self._fx.views != vmax:
                self._fx = DV.FactoredShGrad(vmax)
            fx = self._fx
        # every render of the step carries its own options (no process-wide switch): gradients of further views are
        # added inside the chain-rule kernel, the SH gradient goes to this trainer's own FactoredShGrad
        opts = RenderOptions(mode=self.mode, accumulate=True, sh_sink=fx)
        if fx is not None:     # (rows allocated here, on the caller's stream, before the views fork onto their lanes)
            fx.begin_step(self.params["pws"].shape[0], self.params["pws"].device)
        with _fused.deferred() as d:
            loss_sum, gnorm, count = self._render_views(mine, len(view_ids), opts)
            incomplete = d.commit()
        if incomplete:
            self.redone_steps += 1
            self.opt.zero_grad(set_to_none=True)
            if fx is not None:
                fx.restart()
            loss_sum, gnorm, count = self._render_views(mine, len(view_ids), opts)   # validated render by render
        others = self.params
        sh_rows = None
        if fx is not None:   # (a collective when world > 1; the loss already carries 1 / views: a SUM over ranks)
            if isinstance(self.opt, FusedAdam):
                # the optimizer forms each Gaussian's SH gradient row in LDS (egs_adam_sh_factored): the 4 sh_dim-byte
                # rows are never written or read
                taken = fx.take()
                sh_rows = None if taken is None else (taken[0], 1.0, self.params["pws"], self.params["low_shs"],
                                                      self.params["high_shs"])
            else:
                fx.finish(self.params["pws"], self.params["low_shs"], self.params["high_shs"], average=False)
            others = {k: v for k, v in self.params.items() if k not in ("low_shs", "high_shs")}
        if self.world > 1:   # sum over ranks of (sum over local views)/V == mean over all views
            DV.allreduce_sum_(DV.coalesce_grads(list(others.values())) + [gnorm, count, loss_sum])
        self.grad_accum += gnorm
        self.vis_count += count
        if sh_rows is not None:
            self.opt.step(factored_sh=sh_rows)
        else:
            self.opt.step()
        self.density.update_pws_lr(self.opt)                                     # gsmodel.py:180-183, 332-338
        self.iteration += 1
        mean = loss_sum / len(view_ids)
        return float(mean) if sync else mean

    def densify(self, verbose: bool = False):
        """Prune / clone / split on the statistics gathered since the last call (train.py:71-73 ->
        gsmodel.py:232-317).  Identical on every rank (statistics are already all-reduced)."""
        self.density.set_density_info(self.grad_accum, self.vis_count)
        report = self.density.update_gaussian_density(self.params, self.opt, verbose=verbose)
        n = self.params["pws"].shape[0]
        self.grad_accum = torch.zeros(n, device=self.device)
        self.vis_count = torch.zeros(n, dtype=torch.int32, device=self.device)
        return report

    def reset_alpha(self):
        """train.py:74-76 -> gsmodel.py:319-330."""
        self.density.reset_alpha(self.params, self.opt)

    def fit(self, epochs: int, views_per_step: int = None, rng_seed: int = 0, densify_every: int = 5,
            reset_alpha_every: int = 15, densify_until: int = 50, verbose: bool = False) -> List[float]:
        """The epoch loop of train.py:44-80: shuffled views, ``views_per_step`` views per optimizer step
        (1 in the reference), densification every 5th and alpha reset every 15th epoch in (1, 50]."""
        vps = views_per_step or self.world
        if vps < self.world or len(self.cams) < self.world:
            raise ValueError("views_per_step=%d, %d cameras: every one of the %d ranks needs a view in each step"
                             % (vps, len(self.cams), self.world))
        order_rng = np.random.default_rng(rng_seed)            # same permutation on every rank
        history = []
        for epoch in range(epochs):
            idxs = order_rng.permutation(len(self.cams))
            total, steps = torch.zeros((), device=self.device), 0
            # every view is visited each epoch (train.py:48): the last step of an epoch takes the remaining
            # views when they still give every rank one, otherwise they join the step before it
            cuts = list(range(0, len(idxs), vps))
            if len(cuts) > 1 and len(idxs) - cuts[-1] < self.world:
                cuts.pop()
            for j, i in enumerate(cuts):
                end = cuts[j + 1] if j + 1 < len(cuts) else len(idxs)
                total += self.step([int(v) for v in idxs[i:end]], sync=False)
                steps += 1
            history.append(float(total) / max(steps, 1))
            if verbose and self.rank == 0:
                print("epoch:%d avg_loss:%f" % (epoch, history[-1]))
            if 1 < epoch <= densify_until:
                if epoch % densify_every == 0:
                    self.densify(verbose and self.rank == 0)
                if epoch % reset_alpha_every == 0:
                    self.reset_alpha()
        return history

    def save(self, fn: str) -> np.ndarray:
        """Checkpoint of ACTIVATED parameters, dtype == gau_io.py:7-12 (gau_io.py:141-156)."""
        with torch.no_grad():
            pws, shs, alphas, scales, rots = (x.detach().cpu().numpy() for x in activate(self.params))
        gs = np.rec.fromarrays([pws, rots, scales, alphas.reshape(-1), shs], dtype=gsdata_type(shs.shape[1]))
        np.save(fn, gs)
        return gs
