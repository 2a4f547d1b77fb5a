"""COLMAP sparse-model readers: the counterpart of the reference's
``gsplat/read_write_model.py`` (itself adapted from COLMAP's script).

Binary layouts (little endian), as written by COLMAP's ``Reconstruction::Write*Binary``:

* ``cameras.bin``  : u64 count; per camera  i32 id, i32 model, u64 width, u64 height, f64 params[model]
* ``images.bin``   : u64 count; per image   i32 id, f64 qvec[4] (w,x,y,z), f64 tvec[3], i32 camera_id,
                     NUL-terminated name, u64 n2d, n2d x (f64 x, f64 y, i64 point3D_id)
* ``points3D.bin`` : u64 count; per point   u64 id, f64 xyz[3], u8 rgb[3], f64 error, u64 track,
                     track x (i32 image_id, i32 point2D_idx)

Same return types as the reference (dicts of namedtuples keyed by id, ``Image.qvec2rotmat``), so code
written against it keeps working; records are decoded with ``numpy.frombuffer`` on the whole file
instead of one ``struct.unpack`` per field.
"""
from __future__ import annotations

import collections
import os

import numpy as np

from .scene import gsdata_type

SH_C0_0 = 0.28209479177387814

Camera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])
BaseImage = collections.namedtuple("Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])
Point3D = collections.namedtuple("Point3D", ["id", "xyz", "rgb", "error", "image_ids", "point2D_idxs"])

# model id -> (name, number of parameters)   (read_write_model.py:62-74)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                 4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


def qvec2rotmat(qvec) -> np.ndarray:
    """(w, x, y, z) -> 3x3, no normalisation (read_write_model.py:241-260)."""
    w, x, y, z = qvec
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


class Image(BaseImage):
    def qvec2rotmat(self):
        return qvec2rotmat(self.qvec)


class ColmapFormatError(ValueError):
    pass


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def _take(raw, dtype, count, at, path):
    dtype = np.dtype(dtype)
    end = at + dtype.itemsize * count
    if end > len(raw):
        raise ColmapFormatError("%s: truncated at byte %d" % (path, at))
    return np.frombuffer(raw, dtype=dtype, count=count, offset=at), end


def read_cameras_binary(path) -> dict:
    """read_write_model.py:95-127."""
    raw = _read(path)
    (n,), at = _take(raw, "<u8", 1, 0, path)
    cameras = {}
    for _ in range(int(n)):
        head, at = _take(raw, [("id", "<i4"), ("model", "<i4"), ("w", "<u8"), ("h", "<u8")], 1, at, path)
        cid, model = int(head["id"][0]), int(head["model"][0])
        if model not in CAMERA_MODELS:
            raise ColmapFormatError("%s: unknown camera model id %d" % (path, model))
        name, npar = CAMERA_MODELS[model]
        params, at = _take(raw, "<f8", npar, at, path)
        cameras[cid] = Camera(id=cid, model=name, width=int(head["w"][0]), height=int(head["h"][0]),
                              params=np.array(params))
    return cameras


def read_images_binary(path) -> dict:
    """read_write_model.py:130-176 (dict in file order)."""
    raw = _read(path)
    (n,), at = _take(raw, "<u8", 1, 0, path)
    head_t = np.dtype([("id", "<i4"), ("q", "<f8", (4,)), ("t", "<f8", (3,)), ("cam", "<i4")])
    p2d_t = np.dtype([("xy", "<f8", (2,)), ("p3d", "<i8")])
    images = {}
    for _ in range(int(n)):
        head, at = _take(raw, head_t, 1, at, path)
        end = raw.find(b"\x00", at)
        if end < 0:
            raise ColmapFormatError("%s: unterminated image name" % path)
        name = raw[at:end].decode("utf-8")
        (npt,), at = _take(raw, "<u8", 1, end + 1, path)
        p2d, at = _take(raw, p2d_t, int(npt), at, path)
        iid = int(head["id"][0])
        images[iid] = Image(id=iid, qvec=np.array(head["q"][0]), tvec=np.array(head["t"][0]),
                            camera_id=int(head["cam"][0]), name=name, xys=np.array(p2d["xy"]).reshape(-1, 2),
                            point3D_ids=np.array(p2d["p3d"]).astype(np.int64))
    return images


def read_points3D_binary(path):
    """All points as arrays: (ids u64 [N], xyz f64 [N,3], rgb u8 [N,3], error f64 [N], track_len [N])."""
    raw = _read(path)
    (n,), at = _take(raw, "<u8", 1, 0, path)
    n = int(n)
    head_t = np.dtype([("id", "<u8"), ("xyz", "<f8", (3,)), ("rgb", "u1", (3,)), ("err", "<f8"), ("track", "<u8")])
    assert head_t.itemsize == 51
    out = np.empty(n, dtype=head_t)
    for i in range(n):
        rec, at = _take(raw, head_t, 1, at, path)
        out[i] = rec[0]
        at += 8 * int(rec["track"][0])
        if at > len(raw):
            raise ColmapFormatError("%s: truncated track" % path)
    return out["id"].copy(), out["xyz"].copy(), out["rgb"].copy(), out["err"].copy(), out["track"].astype(np.int64)


def points_to_gaussians(xyz, rgb, nn_sqdist=None) -> np.ndarray:
    """Initial Gaussians from a coloured point cloud (read_write_model.py:197-232): identity rotation,
    alpha 0.8, degree-0 SH from the colour, isotropic scale = clip(d, 0.01, 3) where d is the SQUARED
    distance to the nearest other point (the reference passes faiss's squared L2 on unchanged).
    ``nn_sqdist``: callable points[N,3] -> [N]; default = the HIP kernel (knn.nn_sqdist)."""
    pws = np.asarray(xyz, np.float64).astype(np.float32)
    n = pws.shape[0]
    shs = ((np.asarray(rgb, np.float64) / 255 - 0.5) / SH_C0_0).astype(np.float32)
    rots = np.zeros([n, 4], np.float32)
    rots[:, 0] = 1
    alphas = (np.ones([n]) * 0.8).astype(np.float32)
    if nn_sqdist is None:
        from .knn import nn_sqdist as dev_nn
        d = dev_nn(pws).cpu().numpy() if n else np.zeros(0, np.float32)
    else:
        d = np.asarray(nn_sqdist(pws), np.float32)
    scales = np.clip(d, 0.01, 3)[:, np.newaxis].repeat(3, 1).astype(np.float32)
    return np.rec.fromarrays([pws, rots, scales, alphas, shs], dtype=gsdata_type(3))


def read_points_bin_as_gau(path, nn_sqdist=None) -> np.ndarray:
    """read_write_model.py:179-232."""
    _, xyz, rgb, _, _ = read_points3D_binary(path)
    return points_to_gaussians(xyz, rgb, nn_sqdist)


def read_model(path, ext=".bin"):
    """(cameras, images) of a sparse model directory (read_write_model.py:235-238)."""
    return (read_cameras_binary(os.path.join(path, "cameras.bin")),
            read_images_binary(os.path.join(path, "images.bin")))
