"""k_preprocess_fwd / _bwd on a camera that sees only part of the scene (a COLMAP scene seen from inside: BASELINE configs[4]'s
data is on no box): the bench scene's Gaussians, 300 k of them, through (a) the bench camera (all in view) and (b) a
camera in the middle of the cloud with a short focal length (about half of the Gaussians behind it).  Per-kernel averages
of forward + backward steps (fused path, activated tensors)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, fused, scene as S          # noqa: E402
from easygaussiansplatting_amd.function import Camera, GSFunction      # noqa: E402
from tools.benchlib import parse_report                                # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = S.big_scene(N, 1920, 1080, 48)
sc.scales[:] = sc.scales * 0.25            # (small splats: the inside camera sees the near ones large)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1).clone(), t(sc.scales), t(sc.rots)]
for p in P:
    p.requires_grad_(True)
us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
dl = torch.ones((3, 1080, 1920), device=dev) / (3 * 1920 * 1080)
cams = {"bench camera": sc.cam,
        "inside camera": S.Camera(1920, 1080, 500.0, 500.0, 960.0, 540.0, np.eye(3), np.array([0.0, 0.0, 0.0]))}
for name, c in cams.items():
    cam = Camera.from_scene(c, dev)

    def step():
        for p in P:
            p.grad = None
        us0.grad = None
        with fused.deferred() as d:
            img, mask = GSFunction.apply(*P, us0, cam)
            img.backward(dl)
            d.commit()
        return mask
    for _ in range(6):
        mask = step()
    torch.cuda.synchronize()
    lib.egs_prof_set_filter(None); lib.egs_prof_reset(); lib.egs_prof_enable(1)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    lib.egs_prof_enable(0)
    need = lib.egs_prof_report(None, 0)
    buf = ctypes.create_string_buffer(need + 16)
    lib.egs_prof_report(buf, need + 16)
    rep = parse_report(buf.value.decode())
    lib.egs_prof_reset()
    tb = {k: round(tot / cnt * 1e3, 1) for k, (cnt, tot) in rep.items()}
    print("%-14s in view (depth > 0.2) %.3f  " % (name, float(mask.float().mean())),
          {k: tb[k] for k in ("k_preprocess_fwd", "k_preprocess_bwd", "k_draw", "k_draw_bwd") if k in tb})
