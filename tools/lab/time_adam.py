"""Time FusedAdam.step over the six parameter groups of a 1 M-Gaussian model (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import scene as S
from easygaussiansplatting_amd.optim import FusedAdam, adam_groups
from easygaussiansplatting_amd.trainer import raw_params_from_scene

sc = S.big_scene(1_000_000, 1920, 1080, 48)
raw = raw_params_from_scene(sc, "cuda")
opt = FusedAdam(adam_groups(raw), eps=1e-15)
for p in raw.values():
    p.grad = torch.randn_like(p) * 1e-3
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
n = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    opt.step()
e1.record(); torch.cuda.synchronize()
print("FusedAdam.step, 59 floats x 1 M: %.1f us" % (e0.elapsed_time(e1) / n * 1e3))
