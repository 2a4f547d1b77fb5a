"""Reference module name for easygaussiansplatting_amd.loss (gsplat/pytorch_ssim.py): ``gau_loss`` and ``ssim``
on the fused HIP loss kernels."""
import torch

from easygaussiansplatting_amd.loss import gau_loss, gau_loss_with_grad  # noqa: F401


def ssim(img1, img2, window_size=11, size_average=True):
    """SSIM of two [3,H,W] (or [1,3,H,W]) images with the reference's 11x11 Gaussian window
    (pytorch_ssim.py:49-60); not differentiable -- use ``gau_loss`` for training."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("only window_size=11, size_average=True (what gau_loss uses)")
    a = img1.reshape(-1, *img1.shape[-2:]) if img1.dim() == 4 else img1
    b = img2.reshape(-1, *img2.shape[-2:]) if img2.dim() == 4 else img2
    stats, _ = gau_loss_with_grad(a.detach().contiguous(), b.detach().contiguous(), need_grad=False)
    return stats[2]
