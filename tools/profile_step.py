#!/usr/bin/env python3
"""Small driver for rocprofv3: a few forward+backward steps of the bench
workload (same code path as bench.py's step) so the trace / counter output stays
small.   rocprofv3 --kernel-trace --stats [--pmc ...] -- python tools/profile_step.py"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warm", type=int, default=3,
                help="steps in front, each waited for: the draw stage's path selection (fused._seg_decision) steers by what "
                     "EARLIER renders reported, and a render's report reaches the host with the next render's binning "
                     "stage -- without them a 3-step counter pass measures the first-sight path (round 6: the segment "
                     "kernels on the bench scene), not the steady state")
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--fwd-only", action="store_true")
ap.add_argument("--scene", default="iid", choices=["iid", "skewed", "skewed_reset"],
                help="iid = the bench scene; skewed / skewed_reset = scene.skewed_scene (1.5 M, heavy-tailed; after reset_alpha)")
ap.add_argument("--mode", default="fused", choices=["fused", "ops"], help="GSFunction evaluation (ops = the seven-op surface)")
ap.add_argument("--train", action="store_true", help="whole optimizer step: GSRawFunction + HIP loss + FusedAdam")
ap.add_argument("--factored", action="store_true", help="--train: SH gradient factored, consumed by FusedAdam")
ap.add_argument("--public-pair", action="store_true",
                help="--mode ops: the plain splat / splatB pair an unmodified reference GSFunction calls (no records handle)")
a = ap.parse_args()

import torch
from easygaussiansplatting_amd import scene as S
from easygaussiansplatting_amd.function import Camera, GSFunction, RenderOptions, render

dev = torch.device("cuda", 0)
opts = RenderOptions(mode=a.mode, ops_use_records=not a.public_pair)       # per call, not a process-wide switch
sc = S.big_scene(a.gaussians, a.width, a.height, 48) if a.scene == "iid" else \
    S.skewed_scene(width=a.width, height=a.height, sh_dim=48, reset_alpha=(a.scene == "skewed_reset"))
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = dict(pws=t(sc.pws), shs=t(sc.shs), alphas=t(sc.alphas).reshape(-1, 1).clone(), scales=t(sc.scales), rots=t(sc.rots))
for p in P.values():
    p.requires_grad_(True)
us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
dl = torch.from_numpy(S.normal(1, 77, (3, a.height, a.width)).astype(np.float32)).to(dev) / (3 * a.width * a.height)
if a.train:
    from easygaussiansplatting_amd.function import GSRawFunction
    from easygaussiansplatting_amd.loss import gau_loss
    from easygaussiansplatting_amd.optim import FusedAdam, adam_groups
    from easygaussiansplatting_amd.trainer import raw_params_from_scene
    raw = raw_params_from_scene(sc, dev)
    opt = FusedAdam(adam_groups(raw), eps=1e-15)
    gt = torch.rand((3, a.height, a.width), device=dev)
for it in range(a.warm + a.steps):
    if 0 < it <= a.warm:
        torch.cuda.synchronize()
    if a.train:
        # what Trainer.step does for one view: deferred validation, the SH gradient factored and consumed by FusedAdam
        from easygaussiansplatting_amd import dist_views as DV, fused
        if a.factored and "fx" not in globals():
            fx = DV.FactoredShGrad(1)
        opt.zero_grad(set_to_none=True)
        if a.factored:      # (Trainer._render_views: a persistent `us` leaf, the loss kernels hand over dL/dimage)
            from easygaussiansplatting_amd.loss import gau_loss_with_grad
            if "us_keep" not in globals():
                us_keep = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
            us = us_keep
            us.grad = None
        else:
            us = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
        if a.factored:
            fx.begin_step(sc.n, dev)
        ropts = RenderOptions(accumulate=a.factored, sh_sink=fx if a.factored else None)
        with fused.deferred() as d:
            img, _ = GSRawFunction.apply(raw["pws"], raw["low_shs"], raw["high_shs"], raw["alphas_raw"],
                                         raw["scales_raw"], raw["rots_raw"], us, cam, ropts)
            if a.factored:
                stats, dimg = gau_loss_with_grad(img.detach(), gt, grad_scale=1.0)
                img.backward(dimg)
            else:
                gau_loss(img, gt).backward()
            assert not d.commit()
        if a.factored:
            rows, _w = fx.take()
            opt.step(factored_sh=(rows, 1.0, raw["pws"], raw["low_shs"], raw["high_shs"]))
        else:
            opt.step()
    elif a.fwd_only:
        with torch.no_grad():
            render(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], cam)
    else:
        from easygaussiansplatting_amd import fused
        with fused.deferred() as d:          # the step bench.py times: validation at commit(), not inside forward
            for p in P.values():
                p.grad = None
            us0.grad = None
            img, _ = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam, opts)
            img.backward(dl)
            assert not d.commit()
torch.cuda.synchronize()
print("done")
