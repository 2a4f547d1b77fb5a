import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, fused, scene as S
from tools.benchlib import scene_leg
dev = torch.device("cuda", 0); lib = _lib.load()
for rs in (False, True):
    out = scene_leg("skewed_reset" if rs else "skewed", S.skewed_scene(reset_alpha=rs), dev, lib, 12, None)
    print({k: v for k, v in out.items() if k != "kernels_avg_us"}); print(out["kernels_avg_us"])
