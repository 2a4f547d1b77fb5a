"""Reference module name gsplat/gsmodel.py on the MI355X kernels: ``GSFunction``, ``GSModel`` and
``get_training_params`` with the reference's signatures, so that the reference's ``train.py`` runs as it is.

``GSModel.forward`` is ONE fused autograd node on the raw tensors (``GSRawFunction``); the density methods
are ``easygaussiansplatting_amd.density.DensityControl``."""
import numpy as np
import torch

from easygaussiansplatting_amd.density import DensityControl
from easygaussiansplatting_amd.function import GSFunction, GSRawFunction  # noqa: F401
from gsplat.utils import *  # noqa: F401,F403


def get_training_params(gs):
    """Leaf tensors + Adam groups for a Gaussian record array (the call reference train.py makes): the repo's own
    ``trainer.raw_params_from_scene`` (raw parameterisation, SH split into degree 0 / rest) and
    ``optim.adam_groups`` (the reference's per-group learning rates), exposed under the reference's name."""
    from easygaussiansplatting_amd.optim import adam_groups
    from easygaussiansplatting_amd.scene import Scene
    from easygaussiansplatting_amd.trainer import raw_params_from_scene
    n = gs["pw"].shape[0]
    scene = Scene(gs["pw"], gs["rot"], gs["scale"], gs["alpha"].reshape(n), np.asarray(gs["sh"]).reshape(n, -1), None)
    params = raw_params_from_scene(scene, "cuda", clamp_alpha=False)
    return params, adam_groups(params)


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
