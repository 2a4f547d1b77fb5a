#!/usr/bin/env python3
"""Long-ish training run on a synthetic scene: many views with different patch counts, densification and
alpha resets in the loop -- exercises the enqueue-ahead forward (capacity growth, overflow redo), the
depth-key hint protocol, optimizer surgery and the flat gradient buffer over hundreds of steps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from easygaussiansplatting_amd import fused, gsplatcu as gsc, scene as S
from easygaussiansplatting_amd.function import Camera, render
from easygaussiansplatting_amd.trainer import Trainer

SKEWED = os.environ.get("SCENE") == "skewed"    # scene.skewed_scene at 1080p: the segment kernels, their hint words and
                                                 # expect_long_walks inside a real loop (densify every 2nd, reset every 6th epoch)
n, W, H, views = int(os.environ.get("N", 200000)), 640, 360, 12
if SKEWED:
    sc = S.skewed_scene()
    n, W, H, views = sc.n, sc.cam.width, sc.cam.height, 8
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, views)]
else:
    sc = S.small_scene(n, W, H, 48, seed=1)
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, views, radius=5.0)]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
with torch.no_grad():
    gts = [render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), c)[0] for c in cams]
start = S.skewed_scene() if SKEWED else S.small_scene(n, W, H, 48, seed=1)
start.shs[:, :3] += (0.3 if SKEWED else 0.6) * S.normal(5, 3, (n, 3)).astype(np.float32)
start.pws += (0.004 if SKEWED else 0.01) * S.normal(6, 4, (n, 3)).astype(np.float32)
tr = Trainer(start, cams, gts, max_steps=2000, scene_size=8.0 if SKEWED else 4.0)
if not SKEWED:
    tr.density.grad_threshold = 2e-7
t0 = time.time()
if SKEWED:
    hist = tr.fit(epochs=20, views_per_step=int(os.environ.get("VPS", 1)), densify_every=2, reset_alpha_every=6, densify_until=18)
else:
    hist = tr.fit(epochs=24, views_per_step=int(os.environ.get("VPS", 1)), densify_every=4, reset_alpha_every=12, densify_until=20)
torch.cuda.synchronize()
print("epochs %d, %.1f s, loss %.4f -> %.4f, gaussians %d -> %d, densifications %d" % (
    len(hist), time.time() - t0, hist[0], hist[-1], n, tr.params["pws"].shape[0], tr.density.round))
print("steps rendered twice (a view outgrew the enqueue-ahead buffers):", tr.redone_steps, "of", tr.iteration)
print("patch capacities learnt:", {k: v for k, v in list(fused._ctx(torch.device("cuda", 0)).capacity.items())[:6]},
      "hints", dict(list(gsc._key_bits.items())[:6]))
assert all(np.isfinite(hist)), hist
assert all(torch.isfinite(v).all() for v in tr.params.values())
print("losses per epoch:", [round(h, 4) for h in hist])
print("hint words (longest list, longest walk) by problem size:", {k: fused.seg_hint(torch.device("cuda", 0), k)
                                                                   for k in list(fused._ctx(torch.device("cuda", 0)).seg_hint)[:6]})
print("OK")
