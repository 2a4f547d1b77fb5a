#!/usr/bin/env python3
"""How much GPU time do two INDEPENDENT render steps share when they run on two streams?  An upper bound for what
pipelining the views of a step (the next view's preprocess + binning under this view's draw kernels) could gain:
K steps on one stream against K steps dealt alternately to two streams, two parameter copies."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import fused, scene as S
from easygaussiansplatting_amd.function import Camera, GSFunction

W, H, N = 1920, 1080, 1_000_000
dev = torch.device("cuda", 0)
sc = S.big_scene(N, W, H, 48)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
def params():
    P = dict(pws=t(sc.pws), shs=t(sc.shs), alphas=t(sc.alphas).reshape(-1, 1).clone(), scales=t(sc.scales), rots=t(sc.rots))
    for p in P.values():
        p.requires_grad_(True)
    return P
PS = [params() for _ in range(4)]
dl = torch.from_numpy(S.normal(1, 77, (3, H, W)).astype(np.float32)).to(dev) / (3 * W * H)
torch.cuda.synchronize()

def step(P):
    us = torch.zeros((N, 2), device=dev, requires_grad=True)
    for p in P.values():
        p.grad = None
    img, _ = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us, cam)
    img.backward(dl)

def run(K, streams):
    with fused.deferred() as d:
        for i in range(K):
            s = streams[i % len(streams)]
            with torch.cuda.stream(s):
                step(PS[i % len(streams)])
        torch.cuda.synchronize()
        assert not d.commit()

ST = [torch.cuda.Stream() for _ in range(4)]
for s in ST:
    s.wait_stream(torch.cuda.current_stream())
run(60, ST[:1]); run(60, ST)
for name, ss in (("1 stream", ST[:1]), ("2 streams", ST[:2]), ("3 streams", ST[:3]), ("4 streams", ST), ("1 stream", ST[:1]), ("2 streams", ST[:2]), ("3 streams", ST[:3]), ("4 streams", ST)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(240, ss)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-12s %.4f ms per step" % (name, dt / 240 * 1e3), flush=True)
