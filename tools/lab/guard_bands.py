"""Ad hoc memory-safety check (no GPU sanitizer on this pool): every buffer the host layer allocates for the kernels
(``torch.empty`` in fused.py / gsplatcu.py: records, lists, workspaces, segment states, outputs) gets a 4-KB guard band of
0xA5 bytes on either side; after forward + backward on scenes that exercise every path (iid, ragged image sizes, the
heavy-tailed scene on the segment kernels, the seven-op surface with and without kept states, an overflowing enqueue-ahead
capacity) every band must be untouched."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import fused, gsplatcu as gsc, scene as S      # noqa: E402
from easygaussiansplatting_amd.function import Camera, GSFunction, RenderOptions   # noqa: E402

GUARD = 4096
bands = []          # (label, base uint8 tensor, payload bytes)
real_empty = torch.empty


class Proxy:
    """stands in for the ``torch`` module inside one host-layer module: ``empty`` on the GPU hands out guarded buffers"""

    def __init__(self, label):
        self._label = label

    def __getattr__(self, name):
        return getattr(torch, name)

    def empty(self, *size, dtype=None, device=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        size = tuple(int(s) for s in size)
        dt = dtype or torch.float32
        dv = torch.device(device) if device is not None else torch.device("cpu")
        if dv.type != "cuda" or kw:
            return real_empty(size, dtype=dt, device=dv, **kw)
        nbytes = int(np.prod(size, dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
        pad = (-nbytes) % 256
        base = real_empty(nbytes + pad + 2 * GUARD, dtype=torch.uint8, device=dv)
        base[:GUARD] = 0xA5
        base[GUARD + nbytes:] = 0xA5
        bands.append((self._label, base, nbytes))
        return base[GUARD:GUARD + nbytes].view(dt).reshape(size)


fused.torch = Proxy("fused")
gsc.torch = Proxy("gsplatcu")


def check(tag):
    torch.cuda.synchronize()
    bad = 0
    for label, base, nbytes in bands:
        lo, hi = base[:GUARD], base[GUARD + nbytes:]
        if not (bool((lo == 0xA5).all()) and bool((hi == 0xA5).all())):
            bad += 1
            print("   GUARD HIT", tag, label, "payload bytes", nbytes, "low band touched", int((lo != 0xA5).sum()),
                  "high band touched", int((hi != 0xA5).sum()))
    print("%-52s %5d buffers, %d with a touched guard band" % (tag, len(bands), bad))
    n = len(bands)
    del bands[:]
    return bad


dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
gsc.set_policy("gsplatcu")
total = 0


def step(sc, cam, opts=None, reps=3):
    P = [dev(sc.pws), dev(sc.shs), dev(sc.alphas).reshape(-1, 1), dev(sc.scales), dev(sc.rots)]
    for p in P:
        p.requires_grad_(True)
    us0 = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    dl = dev(S.normal(3, 5, (3, cam.height, cam.width)).astype(np.float32) / (3 * cam.height * cam.width))
    for _ in range(reps):
        for p in P:
            p.grad = None
        img, _ = GSFunction.apply(*P, us0, cam, opts) if opts is not None else GSFunction.apply(*P, us0, cam)
        img.backward(dl)
    assert all(torch.isfinite(p.grad).all() for p in P)


GSFunction.mode = "fused"
for (n, W, H, K) in ((20_000, 200, 120, 48), (5_000, 17, 33, 3), (60_000, 333, 250, 12), (300_000, 1920, 1080, 48)):
    sc = S.small_scene(n, W, H, K, seed=3) if n < 100_000 else S.big_scene(n, W, H, K)
    step(sc, Camera.from_scene(sc.cam))
    total += check("fused %d Gaussians %dx%d" % (n, W, H))
    for o in (RenderOptions(mode="ops"), RenderOptions(mode="ops", ops_use_records=False)):
        step(sc, Camera.from_scene(sc.cam), o)
    total += check("seven ops (handle, public pair) %d %dx%d" % (n, W, H))
# dense long lists on tiny segments: every kind of work item, first sight and with history, speculation on and off
from easygaussiansplatting_amd import _lib          # noqa: E402
import ctypes as C                                  # noqa: E402
lib = _lib.load()
before = (C.c_int * 2)()
_lib.check(lib.egs_seg_config(0, 0, before))
sc = S.small_scene(60_000, 320, 240, 12, seed=5)
sc.scales[:] = sc.scales * 2.2
for reset in (False, True):
    if reset:
        sc.alphas[:] = np.minimum(sc.alphas, 0.01)
    for L, mn in ((64, 64), (128, 200)):
        _lib.check(lib.egs_seg_config(L, mn, None))
        for spec in ("0", "1", "auto"):
            fused.SEGMENTS, fused.SEG_SPECULATE = "1", spec
            step(sc, Camera.from_scene(sc.cam))
            for o in (RenderOptions(mode="ops"), RenderOptions(mode="ops", ops_use_records=False)):
                step(sc, Camera.from_scene(sc.cam), o)
        total += check("segments L=%d min=%d %s" % (L, mn, "reset" if reset else "opaque"))
_lib.check(lib.egs_seg_config(before[0], before[1], None))
fused.SEGMENTS, fused.SEG_SPECULATE = "auto", "auto"
# the heavy-tailed scene at the production setting, both alphas
for reset in (False, True):
    sc = S.skewed_scene(reset_alpha=reset)
    step(sc, Camera.from_scene(sc.cam))
    for o in (RenderOptions(mode="ops"), RenderOptions(mode="ops", ops_use_records=False)):
        step(sc, Camera.from_scene(sc.cam), o)
    total += check("skewed scene%s, fused + seven ops" % (" after reset_alpha" if reset else ""))
# enqueue-ahead capacity far too small: the draw stage runs on truncated lists, then again
sc = S.small_scene(20_000, 200, 120, 48, seed=3)
cam = Camera.from_scene(sc.cam)
step(sc, cam)
ctx = fused._ctx(torch.device("cuda", 0))
for cap in (64, 1000, 30_000):
    ctx.capacity[(sc.n, 200, 120)] = cap
    step(sc, cam, reps=1)
total += check("overflowing enqueue-ahead capacity")
print("TOTAL buffers with a touched guard band:", total)

# ---- the training-loop pieces (loss, FusedAdam, densification row moves, factored SH exchange buffers, k-NN, viewer prep):
# the same proxy on their modules, a Trainer over a densifying run
import importlib                                                    # noqa: E402
mods = []
for name in ("loss", "optim", "density", "dist_views", "knn", "viewer", "trainer", "function"):
    m = importlib.import_module("easygaussiansplatting_amd." + name)
    if hasattr(m, "torch"):
        m.torch = Proxy(name)
        mods.append(m)
from easygaussiansplatting_amd.function import render               # noqa: E402
from easygaussiansplatting_amd.trainer import Trainer               # noqa: E402
sc = S.small_scene(40_000, 320, 180, 48, seed=1)
cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, 4, radius=5.0)]
with torch.no_grad():
    gts = [render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), c)[0] for c in cams]
start = S.small_scene(40_000, 320, 180, 48, seed=1)
start.shs[:, :3] += 0.6 * S.normal(5, 3, (start.n, 3)).astype(np.float32)
tr = Trainer(start, cams, gts, max_steps=500, scene_size=4.0)
tr.density.grad_threshold = 2e-7
for epoch in range(6):
    for v in range(4):
        tr.step([v], sync=False)
    tr.step([0, 1, 2, 3], sync=False)
    if epoch % 2 == 1:
        tr.densify()
    if epoch == 3:
        tr.reset_alpha()
total += check("Trainer: loss, Adam, densify, reset_alpha, 1 and 4 views per step")
from easygaussiansplatting_amd import knn, viewer                    # noqa: E402
pts = dev(S.normal(2, 9, (30_000, 3)))
knn.nearest_sqdist(pts) if hasattr(knn, "nearest_sqdist") else None
total += check("k-NN") if bands else 0
print("TOTAL (with the training-loop pieces):", total)
