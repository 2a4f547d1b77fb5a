#!/usr/bin/env python3
"""cProfile of the host side of bench.py's step with the forced one-rank gradient exchange (EGS_FORCE_EXCHANGE=1)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from easygaussiansplatting_amd import scene as S, dist_views as DV, fused
from easygaussiansplatting_amd.function import Camera, GSFunction
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sc = S.big_scene()
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = dict(pws=t(sc.pws), shs=t(sc.shs), alphas=t(sc.alphas).reshape(-1, 1).clone(), scales=t(sc.scales), rots=t(sc.rots))
for p in P.values(): p.requires_grad_(True)
us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
dl = torch.from_numpy(S.normal(1, 77, (3, 1080, 1920)).astype(np.float32)).to(dev) / (3 * 1920 * 1080)
ex = DV.ChunkedExchange(1)
order = ("pws", "shs", "alphas", "scales", "rots")
def step():
    for p in P.values(): p.grad = None
    us0.grad = None
    with ex.attach():
        img, _ = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam)
        img.backward(dl)
    ex.finish([P[k] for k in order])
for _ in range(30): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
dist.destroy_process_group()
