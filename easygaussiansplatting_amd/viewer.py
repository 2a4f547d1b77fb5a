"""The viewer's per-frame Gaussian preprocess on the device: what the OpenGL compute shader
``viewer/shaders/gau_prep.glsl`` of the reference computes (dispatched by
``viewer/custom_items/gaussian_item.py:264-272``), through ``egs_viewer_prep``.

    prep, depth = gau_prep(gs_data, view_matrix, projection_matrix, (focal_x, focal_y))

``gs_data``: float32 [N, 11 + sh_dim] rows ``{pos 3, rot 4 (w,x,y,z), scale 3, alpha, sh}`` -- the array
``GaussianItem.setData`` uploads (gaussian_item.py:226-241) -- or a record array of ``gsdata_type``;
matrices: 4x4, mathematical convention (``pc = V @ pw``), i.e. what gaussian_item.py holds before
``set_uniform_mat4`` transposes them for OpenGL.  ``prep`` [N,12] = ``{u 3 (NDC), covinv 3, color 3, area 2,
alpha}``; rows the shader culls carry ``u = -100`` and are otherwise zero here (the shader leaves stale
buffer contents).  The OpenGL drawing itself (gau_vert/gau_frag, the bitonic sort) is out of scope.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def pack_gs_data(gs) -> np.ndarray:
    """Record array (``gsdata_type``) -> the interleaved float32 [N, 11 + K] layout of gau_prep.glsl:33-37."""
    n = gs.shape[0]
    sh = np.asarray(gs["sh"], np.float32).reshape(n, -1)
    return np.concatenate([np.asarray(gs["pw"], np.float32), np.asarray(gs["rot"], np.float32),
                           np.asarray(gs["scale"], np.float32), np.asarray(gs["alpha"], np.float32).reshape(n, 1),
                           sh], axis=1)


def gau_prep(gs_data, view_matrix, projection_matrix, focal):
    lib = _lib.load()
    if isinstance(gs_data, np.ndarray) and gs_data.dtype.names:
        gs_data = pack_gs_data(gs_data)
    if not isinstance(gs_data, torch.Tensor):
        gs_data = torch.from_numpy(np.ascontiguousarray(gs_data, np.float32))
    gs_data = gs_data.to("cuda", torch.float32).contiguous()
    if gs_data.dim() != 2 or gs_data.shape[1] - 11 not in (3, 12, 27, 48):
        raise ValueError("gs_data must be [N, 11 + sh_dim] with sh_dim in (3, 12, 27, 48)")
    n, sh_dim = gs_data.shape[0], gs_data.shape[1] - 11
    V = np.ascontiguousarray(np.asarray(view_matrix, np.float32).reshape(4, 4))
    P = np.ascontiguousarray(np.asarray(projection_matrix, np.float32).reshape(4, 4))
    prep = torch.zeros((n, 12), dtype=torch.float32, device=gs_data.device)
    depth = torch.empty((n,), dtype=torch.float32, device=gs_data.device)
    fp = C.POINTER(C.c_float)
    _lib.check(lib.egs_viewer_prep(n, sh_dim, C.c_void_p(gs_data.data_ptr()), V.ctypes.data_as(fp),
                                   P.ctypes.data_as(fp), float(focal[0]), float(focal[1]),
                                   C.c_void_p(prep.data_ptr()), C.c_void_p(depth.data_ptr()),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return prep, depth
