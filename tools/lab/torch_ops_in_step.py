#!/usr/bin/env python3
"""Which ATen ops (torch-side kernels) run inside one fused forward+backward step?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from easygaussiansplatting_amd import scene as S
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
def step():
    for p in P.values():
        p.grad = None
    us0.grad = None
    img, _ = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam)
    img.backward(dl)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
for ev in prof.events():
    if ev.name in ("aten::zeros", "aten::fill_", "aten::to", "aten::copy_"):
        print(ev.name, ev.input_shapes, [str(f) for f in (ev.stack or [])][:6])
