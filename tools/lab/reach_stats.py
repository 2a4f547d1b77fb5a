#!/usr/bin/env python3
"""How tight is the block cull of the draw kernels?  For every (tile, entry) of the bench scene: the 8x8 blocks the
certain-miss BOX of the record reaches (what reach_mask tests) against the blocks in which some point really has
alpha' >= alpha_skip (exact maximum of the concave exponent over the block's rectangle).  The difference is the
number of block evaluations an exact test would save (ignoring finished pixels)."""
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
    img, mask, st = fused.forward(t(sc.pws), t(sc.shs), t(sc.alphas), t(sc.scales), t(sc.rots), cam)
torch.cuda.synchronize()
P = st.patch_count()
ranges = st.ranges.long()
gx = (W + 15) // 16
T = ranges.shape[0]
lens = ranges[:, 1] - ranges[:, 0]
tile_of = torch.repeat_interleave(torch.arange(T, device=dev), lens)          # [P]
g = st.gaussian_ids().long()
rec = st.rec[g]                                                              # [P, 12]
ux, uy, qxx, qxy, qyy = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4]
e1, e2, thr = rec[:, 9], rec[:, 10], rec[:, 11]
tx0 = (tile_of % gx).float() * 16
ty0 = (tile_of // gx).float() * 16
tot_box = tot_exact = tot_pix = 0
never = torch.isinf(thr) & (thr > 0)
for k in range(4):
    x0 = tx0 + 8 * (k & 1); y0 = ty0 + 8 * (k >> 1); x1 = x0 + 7; y1 = y0 + 7
    box = (ux + e1 >= x0) & (ux - e1 <= x1) & (uy + e2 >= y0) & (uy - e2 <= y1) & ~never
    # exact: max of e over the rectangle (concave quadratic): interior point or the four edges
    inside = (ux >= x0) & (ux <= x1) & (uy >= y0) & (uy <= y1)
    best = torch.full_like(ux, -float("inf"))
    for xe in (x0, x1):
        dx = xe - ux
        dy = (-(qxy * dx) / (2 * qyy)).clamp(min=0).mul(0) + torch.minimum(torch.maximum(-(qxy * dx) / (2 * qyy), y0 - uy), y1 - uy)
        best = torch.maximum(best, qxx * dx * dx + qxy * dx * dy + qyy * dy * dy)
    for ye in (y0, y1):
        dy = ye - uy
        dx = torch.minimum(torch.maximum(-(qxy * dy) / (2 * qxx), x0 - ux), x1 - ux)
        best = torch.maximum(best, qxx * dx * dx + qxy * dx * dy + qyy * dy * dy)
    exact = (inside | (best >= thr)) & ~never
    # pixel-exact: does any of the 64 pixel centres pass?
    px = x0[:, None] + torch.arange(8, device=dev).float()[None, :]
    py = y0[:, None] + torch.arange(8, device=dev).float()[None, :]
    hit = torch.zeros_like(box)
    CH = 1 << 20
    for a in range(0, P, CH):
        b = min(P, a + CH)
        dxp = (px[a:b] - ux[a:b, None])[:, None, :]; dyp = (py[a:b] - uy[a:b, None])[:, :, None]
        e = qxx[a:b, None, None] * dxp * dxp + qxy[a:b, None, None] * dxp * dyp + qyy[a:b, None, None] * dyp * dyp
        hit[a:b] = (e >= thr[a:b, None, None]).flatten(1).any(1) & ~never[a:b]
    assert not (hit & ~exact).any() and not (exact & ~box).any(), ((hit & ~exact).sum().item(), (exact & ~box).sum().item())
    tot_box += int(box.sum()); tot_exact += int(exact.sum()); tot_pix += int(hit.sum())
print("P = %d (tile, entry) pairs; blocks reached by the box %.3f per entry, by the exact rectangle test %.3f, "
      "with a pixel centre above the threshold %.3f" % (P, tot_box / P, tot_exact / P, tot_pix / P))
