#!/usr/bin/env python3
"""cProfile of the host side of fused forward+backward steps on a scene small enough for the host to set the pace
(10 k Gaussians, 256x256): where the ~0.4 ms of Python / ctypes / allocator time per step go."""
import cProfile, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import fused, scene as S
from easygaussiansplatting_amd.function import Camera, GSFunction
dev = torch.device("cuda", 0)
sc = S.small_scene(10000, 256, 256, 48, seed=3)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1).clone(), t(sc.scales), t(sc.rots)]
for p in P: p.requires_grad_(True)
us = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
dl = torch.ones((3, 256, 256), device=dev) / (3 * 256 * 256)
def step():
    with fused.deferred() as d:
        for p in P: p.grad = None
        us.grad = None
        img, _ = GSFunction.apply(*P, us, cam)
        img.backward(dl)
        assert not d.commit()
for _ in range(50): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
