#!/usr/bin/env python3
"""Histogram of the 4-bit block masks of the culled lists of the bench scene (what the draw kernels walk):
bit k = 8x8 block k of the 16x16 tile (k&1 = right half, k>>1 = lower half)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import fused, scene as S
from easygaussiansplatting_amd.function import Camera

W, H = 1920, 1080
sc = S.big_scene(1_000_000, W, H, 48)
dev = torch.device("cuda", 0)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
with torch.no_grad():
    img, mask, st = fused.forward(t(sc.pws), t(sc.shs), t(sc.alphas), t(sc.scales), t(sc.rots), cam, need_grad=True)
torch.cuda.synchronize()
m = st.block_masks().long()
P = m.numel()
h = torch.bincount(m, minlength=16).cpu().numpy()
print("P = %d" % P)
for k in range(16):
    print("mask %s  %.4f" % (format(k, "04b"), h[k] / P))
pc = np.array([bin(k).count("1") for k in range(16)])
print("blocks per entry %.3f" % (h * pc).sum().__truediv__(P))
hp = lambda k: ((k & 3) != 0) + ((k >> 2) != 0)            # horizontal pairs (0,1) (2,3) touched
vp = lambda k: ((k & 5) != 0) + ((k & 10) != 0)            # vertical pairs (0,2) (1,3)
print("pair bodies per entry: horizontal pairing %.3f, vertical pairing %.3f" %
      (sum(h[k] * hp(k) for k in range(16)) / P, sum(h[k] * vp(k) for k in range(16)) / P))
both_h = lambda k: ((k & 3) == 3) + ((k >> 2) == 3)
print("pairs with both blocks set per entry (horizontal) %.3f" % (sum(h[k] * both_h(k) for k in range(16)) / P))
