"""Test helper: write a tiny COLMAP sparse model + images for a synthetic scene (layouts of
COLMAP's Reconstruction::Write*Binary, see easygaussiansplatting_amd/colmap.py)."""
import os
import struct

import numpy as np


def rotmat2qvec(R):
    from easygaussiansplatting_amd.gau_io import matrix_to_quaternion
    return matrix_to_quaternion(np.asarray(R, np.float64)[None])[0]


def write_scene(root, cams, images_u8, xyz, rgb, model="PINHOLE"):
    """cams: scene.Camera list (all same intrinsics); images_u8: list of [H,W,3] uint8."""
    from PIL import Image
    sparse = os.path.join(root, "sparse", "0")
    os.makedirs(sparse)
    os.makedirs(os.path.join(root, "images"))
    c = cams[0]
    with open(os.path.join(sparse, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", 1))
        if model == "PINHOLE":
            f.write(struct.pack("<iiQQ", 1, 1, c.width, c.height) + struct.pack("<4d", c.fx, c.fy, c.cx, c.cy))
        else:
            f.write(struct.pack("<iiQQ", 1, 0, c.width, c.height) + struct.pack("<3d", c.fx, c.cx, c.cy))
    with open(os.path.join(sparse, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for i, (cam, img) in enumerate(zip(cams, images_u8)):
            name = "view_%02d.png" % i
            Image.fromarray(img).save(os.path.join(root, "images", name))
            q = rotmat2qvec(cam.Rcw)
            f.write(struct.pack("<idddddddi", i + 1, *q, *np.asarray(cam.tcw, np.float64), 1))
            f.write(name.encode() + b"\x00" + struct.pack("<Q", 0))
    with open(os.path.join(sparse, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(xyz)))
        for i in range(len(xyz)):
            f.write(struct.pack("<QdddBBBd", i + 1, *[float(v) for v in xyz[i]], *[int(v) for v in rgb[i]], 0.5))
            f.write(struct.pack("<Q", 0))
