"""Time the fused HIP loss (k_ssim_fwd + k_loss_finalize + k_ssim_bwd) at 1920x1080 with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd.loss import gau_loss_with_grad

dev = torch.device("cuda", 0)
torch.manual_seed(0)
img = torch.rand(3, 1080, 1920, device=dev); gt = torch.rand(3, 1080, 1920, device=dev)
for _ in range(5):
    gau_loss_with_grad(img, gt)
torch.cuda.synchronize()
n = 100
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    stats, grad = gau_loss_with_grad(img, gt)
e1.record(); torch.cuda.synchronize()
print("hip gau_loss fwd+grad at 1920x1080: %.1f us  loss %.6f  |grad|sum %.6e" %
      (e0.elapsed_time(e1) / n * 1e3, float(stats[0]), float(grad.abs().sum())))
