#!/usr/bin/env python3
"""The seven-op training step as an UNMODIFIED reference GSFunction runs it (public splat / splatB, no handle), with the
content-validated keeping on and off, and this package's handle form -- same process, interleaved rounds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import gsplatcu as gsc, scene as S
from easygaussiansplatting_amd.function import Camera, GSFunction

sc = S.big_scene()
dev = torch.device("cuda", 0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1), t(sc.scales), t(sc.rots)]
for p in P:
    p.requires_grad_(True)
us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
cam = Camera.from_scene(sc.cam, dev)
dl = t(S.normal(1, 77, (3, sc.cam.height, sc.cam.width))) / (3 * sc.cam.height * sc.cam.width)
GSFunction.mode = "ops"


def step():
    for p in P:
        p.grad = None
    us0.grad = None
    img, _ = GSFunction.apply(*P, us0, cam)
    img.backward(dl)


def timed(n=100):
    for _ in range(20):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(300):
    step()
for rnd in range(3):
    GSFunction.ops_use_records = True
    a = timed()
    GSFunction.ops_use_records = False
    gsc.set_memo(True); b = timed()
    gsc.set_memo(False); c = timed()
    gsc.set_memo(True)
    print("round %d: handle %.4f ms   public pair, validated keeping %.4f ms   public pair, nothing kept %.4f ms" % (rnd, a, b, c), flush=True)
