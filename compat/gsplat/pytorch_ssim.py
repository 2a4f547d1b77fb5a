"""Reference module name for easygaussiansplatting_amd.loss (gsplat/pytorch_ssim.py): ``gau_loss`` and ``ssim``
on the fused HIP loss kernels."""
import torch

from easygaussiansplatting_amd.loss import gau_loss, gau_loss_with_grad  # noqa: F401


def ssim(img1, img2, window_size=11, size_average=True):
    """SSIM with the reference's 11x11 Gaussian window (pytorch_ssim.py:49-60) of two [3,H,W] images or two
    [B,3,H,W] batches: the mean over everything (``size_average=True``, a 0-dim tensor) or one mean per batch
    element (``False``, shape [B], pytorch_ssim.py:46-47).  Not differentiable -- use ``gau_loss`` for training."""
    if window_size != 11:
        raise NotImplementedError("the fused kernel implements the 11x11 window gau_loss uses")
    a = img1 if img1.dim() == 4 else img1[None]
    b = img2 if img2.dim() == 4 else img2[None]
    if a.shape != b.shape or a.shape[1] != 3:
        raise ValueError("ssim expects two [3,H,W] or [B,3,H,W] tensors of one shape")
    per = torch.stack([gau_loss_with_grad(x.detach().contiguous(), y.detach().contiguous(), need_grad=False)[0][2]
                       for x, y in zip(a, b)])
    return per.mean() if size_average else per
