#!/usr/bin/env python3
"""Digest of one forward (+ backward) of the bench scene and of the 10 k scene: image bytes (sha1), contrib / final_tau
bytes, and the gradient sums -- to compare two builds of the library (python tools/lab/render_digest.py > a.txt)."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import fused, scene as S
from easygaussiansplatting_amd.function import Camera, GSFunction
dev = torch.device("cuda", 0)
h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
for name, sc, W, H in (("1M", S.big_scene(1_000_000, 1920, 1080, 48), 1920, 1080), ("10k", S.small_scene(10000, 256, 256, 48, seed=3), 256, 256),
                       ("20k", S.small_scene(20000, 640, 368, 48, seed=5), 640, 368)):
    cam = Camera.from_scene(sc.cam, dev)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1).clone(), t(sc.scales), t(sc.rots)]
    for p in P: p.requires_grad_(True)
    us = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
    dl = torch.from_numpy(S.normal(1, 77, (3, H, W)).astype(np.float32)).to(dev) / (3 * W * H)
    img, _ = GSFunction.apply(*P, us, cam)
    img.backward(dl)
    torch.cuda.synchronize()
    with torch.no_grad():
        _, _, st = fused.forward(*[p.detach() for p in P], cam)
    torch.cuda.synchronize()
    print(name, "image", h(img), "contrib", h(st.contrib), "tau", h(st.final_tau), "grads",
          " ".join("%.9e" % float(p.grad.double().abs().sum()) for p in P))
