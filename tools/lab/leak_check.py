"""Ad hoc: does anything grow over a long run with many densifications?  scene.small_scene, 8 views, 60 epochs, a densification
every 2nd epoch and a reset_alpha every 7th: device memory in use / reserved and the sizes of the host-side tables (patch
capacities, depth-key hints, hint slots, walk words, tile-order cache, pad index) every 10 epochs."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import fused, gsplatcu as gsc, scene as S   # noqa: E402
from easygaussiansplatting_amd.function import Camera, render              # noqa: E402
from easygaussiansplatting_amd.trainer import Trainer                      # noqa: E402

n, W, H, views = 150_000, 640, 360, 8
sc = S.small_scene(n, W, H, 48, seed=1)
cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, views, radius=5.0)]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
with torch.no_grad():
    gts = [render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), c)[0] for c in cams]
start = S.small_scene(n, W, H, 48, seed=1)
start.shs[:, :3] += 0.6 * S.normal(5, 3, (n, 3)).astype(np.float32)
tr = Trainer(start, cams, gts, max_steps=5000, scene_size=4.0)
tr.density.grad_threshold = 2e-7
ctx = fused._ctx(torch.device("cuda", 0))
rng = np.random.default_rng(0)
for epoch in range(60):
    for v in rng.permutation(views):
        tr.step([int(v)], sync=False)
    if epoch % 2 == 1:
        tr.densify()
    if epoch % 7 == 6:
        tr.reset_alpha()
    if epoch % 10 == 9:
        torch.cuda.synchronize()
        print("epoch %2d  N %7d  in use %5d MiB  reserved %5d MiB  capacity keys %3d  key-bit hints %3d  hint slots %2d  "
              "walk words %2d  tile-order entries %2d  pad index %d  mailbox free %d  pending %d"
              % (epoch, tr.params["pws"].shape[0], torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20,
                 len(ctx.capacity), len(gsc._key_bits), len(ctx.seg_hint), len(ctx.walk_word), len(ctx.tile_work),
                 len(fused._pad_index), len(ctx.free), len(ctx.pending)))
assert all(torch.isfinite(v).all() for v in tr.params.values())
print("OK, redone steps", tr.redone_steps, "of", tr.iteration)
