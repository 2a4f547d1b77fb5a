"""How long does the reference-style loss (0.8 L1 + 0.2 (1-SSIM), 11x11 Gaussian window via five depthwise
conv2d) take in plain PyTorch-ROCm at 1920x1080?  (sizing SURVEY §8f-2)"""
import time
import torch
import torch.nn.functional as F

def window(ws=11, sigma=1.5, ch=3, dev="cuda"):
    x = torch.arange(ws, dtype=torch.float32) - ws // 2
    g = torch.exp(-x * x / (2 * sigma * sigma)); g = g / g.sum()
    return (g[:, None] @ g[None, :]).expand(ch, 1, ws, ws).contiguous().to(dev)

def loss_fn(img, gt, w, lam=0.2):
    c = img.shape[-3]
    conv = lambda t: F.conv2d(t, w, padding=w.shape[-1] // 2, groups=c)
    mu1, mu2 = conv(img), conv(gt)
    s11 = conv(img * img) - mu1 * mu1; s22 = conv(gt * gt) - mu2 * mu2; s12 = conv(img * gt) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return (1 - lam) * (img - gt).abs().mean() + lam * (1 - ssim.mean())

dev = "cuda"
img = torch.rand(3, 1080, 1920, device=dev, requires_grad=True); gt = torch.rand(3, 1080, 1920, device=dev)
w = window()
for _ in range(3):
    l = loss_fn(img, gt, w); l.backward(); img.grad = None
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n):
    l = loss_fn(img, gt, w); l.backward(); img.grad = None
torch.cuda.synchronize()
print("torch gau_loss fwd+bwd at 1920x1080: %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
