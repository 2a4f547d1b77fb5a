#!/usr/bin/env python3
"""How much host time does one fused forward+backward step need?  (wall time minus the time blocked in the
read-back of P and in the final synchronise)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import scene as S, gsplatcu as gsc
from easygaussiansplatting_amd.function import Camera, GSFunction
dev = torch.device("cuda", 0)
sc = S.big_scene(1_000_000, 1920, 1080, 48)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = dict(pws=t(sc.pws), shs=t(sc.shs), alphas=t(sc.alphas).reshape(-1, 1).clone(), scales=t(sc.scales), rots=t(sc.rots))
for p in P.values():
    p.requires_grad_(True)
us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
dl = torch.from_numpy(S.normal(1, 77, (3, 1080, 1920)).astype(np.float32)).to(dev) / (3 * 1920 * 1080)
blocked = [0.0]
orig = torch.Tensor.tolist
def timed_tolist(self):
    t0 = time.perf_counter(); r = orig(self); blocked[0] += time.perf_counter() - t0; return r
torch.Tensor.tolist = timed_tolist
def step():
    for p in P.values():
        p.grad = None
    us0.grad = None
    img, _ = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam)
    img.backward(dl)
for _ in range(5):
    step()
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
n = 50
blocked[0] = 0.0
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("wall %.3f ms/step, blocked in read-back %.3f ms/step, host busy %.3f ms/step, final drain %.3f ms" % (
    (t2 - t0) / n * 1e3, blocked[0] / n * 1e3, (t1 - t0 - blocked[0]) / n * 1e3, (t2 - t1) * 1e3))
