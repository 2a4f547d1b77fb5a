"""COLMAP scene as a training set: the counterpart of ``gsplat/gausplat_dataset.py``.

``GSplatDataset(path)`` expects ``path/sparse/0/{cameras,images,points3D}.bin`` and ``path/images/*``;
item ``i`` is ``(Camera, image[3,H,W] float32 in [0,1] on the device)``; ``.gs`` holds the initial
Gaussians (cached as ``points3D.npy`` next to the model, like the reference) and ``.sence_size``
(sic) = 1.1 x the largest camera-centre distance from the mean centre (gausplat_dataset.py:67-69).
PIL decodes the images (the reference goes through torchvision's ``to_tensor``: uint8 HWC -> float
CHW / 255).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import colmap
from .function import Camera as RenderCamera


class Camera(RenderCamera):
    """Render camera + the bookkeeping fields of gausplat_dataset.py:14-26."""

    def __init__(self, id, width, height, fx, fy, cx, cy, Rcw, tcw, path=None, device="cuda"):
        super().__init__(width, height, fx, fy, cx, cy, Rcw, tcw, device=device, id=id, path=path)


def _to_tensor(img, device):
    a = np.array(img.convert("RGB"), dtype=np.uint8)   # (a writable copy: torch.from_numpy warns otherwise)
    return torch.from_numpy(a).to(device).permute(2, 0, 1).to(torch.float32).div_(255.0).contiguous()


class GSplatDataset(torch.utils.data.Dataset):
    def __init__(self, path, resize_rate=1, device="cuda") -> None:
        super().__init__()
        from PIL import Image
        self.device = device
        self.resize_rate = resize_rate
        sparse = os.path.join(path, "sparse", "0")
        camera_params, image_params = colmap.read_model(sparse, ext=".bin")
        self.cameras, self.images = [], []
        for ip in image_params.values():
            cp = camera_params[ip.camera_id]
            im_path = os.path.join(path, "images", ip.name)
            image = Image.open(im_path)
            if resize_rate != 1:
                image = image.resize((int(image.width * resize_rate), int(image.height * resize_rate)))
            w_scale, h_scale = image.width / cp.width, image.height / cp.height
            # the reference reads params[0..3] as fx, fy, cx, cy, i.e. it assumes a PINHOLE-like model
            # (gausplat_dataset.py:50-53); SIMPLE_* models store (f, cx, cy)
            if cp.model in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL", "SIMPLE_RADIAL_FISHEYE", "RADIAL", "RADIAL_FISHEYE"):
                f, cx, cy = cp.params[0], cp.params[1], cp.params[2]
                fx, fy = f * w_scale, f * h_scale
                cx, cy = cx * w_scale, cy * h_scale
            else:
                fx, fy = cp.params[0] * w_scale, cp.params[1] * h_scale
                cx, cy = cp.params[2] * w_scale, cp.params[3] * h_scale
            Rcw = torch.from_numpy(ip.qvec2rotmat()).to(device).to(torch.float32)
            tcw = torch.from_numpy(np.asarray(ip.tvec)).to(device).to(torch.float32)
            self.cameras.append(Camera(ip.id, image.width, image.height, float(fx), float(fy), float(cx), float(cy),
                                       Rcw, tcw, im_path, device))
            self.images.append(_to_tensor(image, device))
        cache = os.path.join(sparse, "points3D.npy")
        if os.path.exists(cache):
            self.gs = np.load(cache)
        else:
            self.gs = colmap.read_points_bin_as_gau(os.path.join(sparse, "points3D.bin"))
            try:
                np.save(cache, self.gs)
            except OSError:
                pass                                   # read-only dataset directory
        twcs = torch.stack([c.twc for c in self.cameras])
        cam_dist = torch.linalg.norm(twcs - torch.mean(twcs, dim=0), dim=1)
        self.sence_size = float(torch.max(cam_dist)) * 1.1
        self.scene_size = self.sence_size

    def __getitem__(self, index: int):
        return self.cameras[index], self.images[index]

    def __len__(self) -> int:
        return len(self.images)
