"""Reference module name gsplat/gsmodel.py on the MI355X kernels: ``GSFunction``, ``GSModel`` and
``get_training_params`` with the reference's signatures, so that the reference's ``train.py`` runs as it is.

``GSModel.forward`` is ONE fused autograd node on the raw tensors (``GSRawFunction``); the density methods
are ``easygaussiansplatting_amd.density.DensityControl``."""
import numpy as np
import torch

from easygaussiansplatting_amd.density import DensityControl
from easygaussiansplatting_amd.function import GSFunction, GSRawFunction  # noqa: F401
from gsplat.utils import *  # noqa: F401,F403
from gsplat.utils import get_alphas_raw, get_scales_raw


def get_training_params(gs):
    """gsmodel.py:96-129: raw leaf tensors + the per-group Adam settings."""
    dev = "cuda"
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).type(torch.float32).to(dev)
    pws = f(gs["pw"]).requires_grad_()
    rots_raw = f(gs["rot"]).requires_grad_()
    scales_raw = get_scales_raw(f(gs["scale"])).requires_grad_()
    alphas_raw = get_alphas_raw(f(gs["alpha"][:, np.newaxis])).requires_grad_()
    shs = f(gs["sh"]).reshape(pws.shape[0], -1)
    low_shs = shs[:, :3].contiguous()
    high_shs = torch.ones_like(low_shs).repeat(1, 15) * 0.001
    high_shs[:, :shs[:, 3:].shape[1]] = shs[:, 3:]
    low_shs = low_shs.requires_grad_()
    high_shs = high_shs.requires_grad_()
    params = {"pws": pws, "low_shs": low_shs, "high_shs": high_shs, "alphas_raw": alphas_raw,
              "scales_raw": scales_raw, "rots_raw": rots_raw}
    adam_params = [{"params": [params["pws"]], "lr": 0.001, "name": "pws"},
                   {"params": [params["low_shs"]], "lr": 0.001, "name": "low_shs"},
                   {"params": [params["high_shs"]], "lr": 0.001 / 20, "name": "high_shs"},
                   {"params": [params["alphas_raw"]], "lr": 0.05, "name": "alphas_raw"},
                   {"params": [params["scales_raw"]], "lr": 0.005, "name": "scales_raw"},
                   {"params": [params["rots_raw"]], "lr": 0.001, "name": "rots_raw"}]
    return params, adam_params


class GSModel(torch.nn.Module, DensityControl):
    """gsmodel.py:169-338."""

    def __init__(self, sense_size, max_steps):
        torch.nn.Module.__init__(self)
        DensityControl.__init__(self, sense_size, max_steps)
        self.cam = None

    def forward(self, pws, low_shs, high_shs, alphas_raw, scales_raw, rots_raw, cam):
        self.cam = cam
        # us is not involved in the forward pass; it collects dloss_dus for the density statistics
        self.us = torch.zeros([pws.shape[0], 2], dtype=torch.float32, device=pws.device, requires_grad=True)
        image, self.mask = GSRawFunction.apply(pws, low_shs, high_shs, alphas_raw, scales_raw, rots_raw, self.us, cam)
        return image

    def update_density_info(self):
        """Do it after backward (gsmodel.py:214-230)."""
        DensityControl.update_density_info(self, self.us.grad, self.mask)
        del self.us.grad
        del self.mask

    def update_gaussian_density(self, params, optimizer):
        report = DensityControl.update_gaussian_density(self, params, optimizer)
        print("---------------------")
        print("gaussian density update report")
        print("pruned num: ", report["pruned"])
        print("cloned num: ", report["cloned"])
        print("splited num: ", report["splited"])
        print("total gaussian number: ", report["total"])
        print("---------------------")
