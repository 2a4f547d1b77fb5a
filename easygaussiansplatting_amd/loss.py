"""Fused training loss: the counterpart of the reference's ``gau_loss``
(gsplat/pytorch_ssim.py:63-66: 0.8 * L1 + 0.2 * (1 - SSIM), 11x11 Gaussian window).

Same call, same value, same gradient; underneath two tiled HIP kernels
(csrc/egs_loss.hip) that also produce dloss/dimage in the forward pass, so that
``loss.backward()`` hands splatB its ``dloss_dgammas`` without running five
depthwise conv2d backward passes (10.9 ms -> see DESIGN.md at 1920x1080).

    from easygaussiansplatting_amd.loss import gau_loss
    loss = gau_loss(image, gt_image)          # image: GSFunction output [3,H,W]
    loss.backward()
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .gsplatcu import _chk, _lib_on, _ptr, _stream


def gau_loss_with_grad(image, gt_image, loss_lambda=0.2, need_grad=True, grad_scale=1.0):
    """-> (stats[3] = {loss, l1, ssim} device tensor, grad_scale * dloss_dimage [3,H,W] or None).
    A training loop that knows the factor in front of the loss (1 / views) passes it here and calls
    ``image.backward(grad)``: no autograd node for the loss, no ones-fill, no scaling pass over the image."""
    image = _chk(image, "image", torch.float32, (3, None, None))
    H, W = int(image.shape[1]), int(image.shape[2])
    gt_image = _chk(gt_image, "gt_image", torch.float32, (3, H, W))
    lib = _lib_on(image)
    dev = image.device
    stats = torch.empty(3, dtype=torch.float32, device=dev)
    grad = torch.empty_like(image) if need_grad else None
    ws_bytes = lib.egs_gau_loss_ws_bytes(H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.egs_gau_loss(H, W, _ptr(image), _ptr(gt_image), float(loss_lambda), float(grad_scale), _ptr(ws),
                                ws_bytes, _ptr(stats), _ptr(grad), _stream()))
    return stats, grad


class _GauLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt_image, loss_lambda):
        stats, grad = gau_loss_with_grad(image.detach(), gt_image.detach(), loss_lambda, image.requires_grad)
        ctx.save_for_backward(grad)
        return stats[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad is not None else None), None, None


def gau_loss(image, gt_image, loss_lambda=0.2):
    """Drop-in for reference gsplat/pytorch_ssim.py:63-66 (scalar tensor, differentiable in ``image``)."""
    return _GauLoss.apply(image, gt_image, loss_lambda)
