#!/usr/bin/env python3
"""How much of each tile list is actually needed per 8x8 block? (sizing early-exit gains)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import scene as S
from easygaussiansplatting_amd.function import Camera, render
dev = torch.device("cuda", 0)
sc = S.big_scene()
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
img, contrib, tau, ranges, gsid = render(t(sc.pws), t(sc.shs), t(sc.alphas), t(sc.scales), t(sc.rots), cam)
H, W = contrib.shape
gy, gx = (H + 15) // 16, (W + 15) // 16
c = torch.zeros((gy * 16, gx * 16), dtype=torch.int32, device=dev); c[:H, :W] = contrib
lens = (ranges[:, 1] - ranges[:, 0]).reshape(gy, gx).float()
blk = c.reshape(gy, 2, 8, gx, 2, 8).permute(0, 3, 1, 4, 2, 5).reshape(gy, gx, 4, 64)
blkmax = blk.max(-1).values.float()            # [gy,gx,4]
tilemax = blkmax.max(-1).values
done = (tau < 1e-4)
print("tiles", gy * gx, "mean len %.1f" % lens.mean().item(), "mean tile maxcont %.1f" % tilemax.mean().item(),
      "mean block maxcont %.1f" % blkmax.mean().item(), "mean pixel contrib %.1f" % c.float().mean().item())
print("fraction of pixels saturated (tau<1e-4): %.3f" % done.float().mean().item())
print("fwd work now  (sum tile_maxcont*4 blocks)  = %.3e" % (tilemax.sum().item() * 4))
print("fwd work with per-block exit (sum blkmax)   = %.3e" % blkmax.sum().item())
print("list total*4 = %.3e" % (lens.sum().item() * 4))
# per-tile work estimates for the scheduling simulation (tools: /tmp/sim.py style): list length, largest contrib,
# sum over the four blocks of their largest contrib (what k_draw_bwd actually walks per block)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "tile_work.npz")
np.savez_compressed(out, lens=lens.cpu().numpy(), tilemax=tilemax.cpu().numpy(), blksum=blkmax.sum(-1).cpu().numpy())
r = (tilemax / lens.clamp(min=1)).flatten()
print("tile maxcont / len: mean %.3f  p10 %.3f  p50 %.3f  p90 %.3f" % (r.mean().item(), r.quantile(0.1).item(),
      r.quantile(0.5).item(), r.quantile(0.9).item()))
print("saved", out)
