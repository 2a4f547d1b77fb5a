"""Nearest-neighbour distances on the device (replaces the reference's faiss call,
gsplat/read_write_model.py:216-220)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def nn_sqdist(points) -> torch.Tensor:
    """[N] float32: squared distance from every point to its nearest OTHER point
    (== faiss.IndexFlatL2(3).search(points, 2)[0][:, 1]).  ``points``: [N,3] tensor or array."""
    lib = _lib.load()
    if not isinstance(points, torch.Tensor):
        points = torch.from_numpy(np.ascontiguousarray(points, np.float32))
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points must be [N, 3]")
    pts = points.to("cuda", torch.float32).contiguous()
    n = pts.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    ws = torch.empty(lib.egs_nn_sqdist_ws_bytes(n), dtype=torch.uint8, device=pts.device)
    _lib.check(lib.egs_nn_sqdist(n, pts.data_ptr(), ws.data_ptr(), ws.numel(), out.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream))
    return out
