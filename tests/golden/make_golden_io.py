#!/usr/bin/env python3
"""Generate fixture G9 (IO / formats) by IMPORTING the reference's
``gsplat/gau_io.py`` and ``gsplat/read_write_model.py`` in the build container.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_io.py

The fixture holds small input FILES as byte strings (a 3DGS ``.ply``, COLMAP
``cameras.bin`` / ``images.bin`` / ``points3D.bin``; all synthesised here with
``struct`` / ``numpy.tobytes`` from seeded values, independent of the repo's own
reader/writer) and what the reference's functions return for them.

Two third-party modules the reference imports are absent from this image; both are
replaced, for this script only, by the thinnest possible equivalents of their
documented behaviour, so that the reference's OWN code (activations, SH
de-interleave, record assembly, clipping of the neighbour distance...) is what
produces every expected value:
* ``plyfile.PlyData.read``  -> header parse + ``numpy.frombuffer`` of the vertex element
  (plyfile's documented behaviour for a binary_little_endian file), with ``elements[0][name]``
  and ``len(elements[0][0])`` as gau_io.py:60-89 uses them;
* ``faiss.IndexFlatL2``     -> exact squared-L2 brute force, results ascending (the published
  contract of IndexFlatL2.search), as read_write_model.py:216-220 uses it.
"""
import io
import os
import struct
import sys
import tempfile
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

import numpy as np  # noqa: E402


# ---- stand-ins for the two absent third-party modules (see module docstring) ------------------------
class _Element:
    def __init__(self, arr):
        self.data = arr

    def __getitem__(self, key):
        return self.data[key]


class PlyData:
    def __init__(self, elements):
        self.elements = elements

    @staticmethod
    def read(path):
        raw = open(path, "rb").read()
        end = raw.index(b"end_header\n") + len(b"end_header\n")
        lines = raw[:end].decode("ascii").split("\n")
        assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
        count, props = None, []
        for ln in lines[2:]:
            t = ln.split()
            if t[:2] == ["element", "vertex"]:
                count = int(t[2])
            elif t and t[0] == "property":
                assert t[1] == "float"
                props.append(t[2])
        arr = np.frombuffer(raw, dtype=[(p, "<f4") for p in props], count=count, offset=end)
        return PlyData([_Element(arr)])


class IndexFlatL2:
    def __init__(self, d):
        self.d = d
        self.x = np.zeros((0, d), np.float32)

    def add(self, x):
        self.x = np.concatenate([self.x, np.asarray(x, np.float32)])

    def search(self, q, k):
        q = np.asarray(q, np.float32)
        d2 = ((q[:, None, :].astype(np.float64) - self.x[None].astype(np.float64)) ** 2).sum(-1)
        idx = np.argsort(d2, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(d2, idx, 1).astype(np.float32), idx


m = types.ModuleType("plyfile"); m.PlyData = PlyData; sys.modules["plyfile"] = m
m = types.ModuleType("faiss"); m.IndexFlatL2 = IndexFlatL2; sys.modules["faiss"] = m
sys.modules["gsplatcu"] = types.ModuleType("gsplatcu")
sys.path.insert(0, REF)

import gsplat.gau_io as ref_io  # noqa: E402
import gsplat.read_write_model as ref_rw  # noqa: E402
sys.path.insert(1, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.golden import _recipe  # noqa: E402

_recipe.assert_reference(ref_io, ref_rw)


# ---- input files ---------------------------------------------------------------------------------
def make_ply(rng, n, sh_rest):
    """A 3DGS point_cloud.ply: x y z nx ny nz f_dc_0..2 f_rest_0..R-1 opacity scale_0..2 rot_0..3."""
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += ["f_rest_%d" % i for i in range(sh_rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    rows = rng.normal(0, 1, (n, len(names))).astype("<f4")
    rows[:, 3:6] = 0
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    hdr += "".join("property float %s\n" % k for k in names) + "end_header\n"
    return hdr.encode("ascii") + rows.tobytes()


def make_colmap(rng, n_cam, n_img, n_pts):
    cams = io.BytesIO()
    cams.write(struct.pack("<Q", n_cam))
    for c in range(n_cam):
        model = (1, 0, 4)[c % 3]                     # PINHOLE, SIMPLE_PINHOLE, OPENCV
        npar = {1: 4, 0: 3, 4: 8}[model]
        cams.write(struct.pack("<iiQQ", c + 1, model, 640 + 32 * c, 480 + 16 * c))
        cams.write(struct.pack("<%dd" % npar, *rng.uniform(100, 700, npar)))
    imgs = io.BytesIO()
    imgs.write(struct.pack("<Q", n_img))
    for i in range(n_img):
        q = rng.normal(0, 1, 4); q /= np.linalg.norm(q)
        t = rng.normal(0, 2, 3)
        imgs.write(struct.pack("<idddddddi", 10 + i, *q, *t, 1 + i % n_cam))
        imgs.write(("frame_%03d.jpg" % i).encode("utf-8") + b"\x00")
        npt = int(rng.integers(0, 6))
        imgs.write(struct.pack("<Q", npt))
        for _ in range(npt):
            imgs.write(struct.pack("<ddq", *rng.uniform(0, 600, 2), int(rng.integers(-1, 50))))
    pts = io.BytesIO()
    pts.write(struct.pack("<Q", n_pts))
    xyz = rng.normal(0, 1.5, (n_pts, 3))
    xyz[7] = xyz[3]                                  # an exact duplicate: neighbour distance 0 -> clipped to 0.01
    xyz[11] = xyz[12] + 40.0                         # an outlier: distance clipped to 3
    for i in range(n_pts):
        rgb = rng.integers(0, 256, 3)
        pts.write(struct.pack("<QdddBBBd", 100 + i, *xyz[i], *[int(v) for v in rgb], float(rng.uniform(0, 2))))
        tl = int(rng.integers(0, 5))
        pts.write(struct.pack("<Q", tl))
        pts.write(struct.pack("<%di" % (2 * tl), *[int(v) for v in rng.integers(0, 100, 2 * tl)]))
    return cams.getvalue(), imgs.getvalue(), pts.getvalue()


def main():
    rng = np.random.default_rng(9)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # (a file without f_rest_* makes gau_io.py:91 raise -- reshape(-1, 3, 0) -- so degree 0 has no golden)
        for tag, n, rest in (("deg3", 37, 45), ("deg2", 6, 24), ("deg1", 5, 9)):
            raw = make_ply(rng, n, rest)
            fn = os.path.join(tmp, tag + ".ply")
            open(fn, "wb").write(raw)
            gs = ref_io.load_ply(fn)                                  # gau_io.py:60-105
            out["ply_%s_bytes" % tag] = np.frombuffer(raw, np.uint8)
            for f in ("pw", "rot", "scale", "alpha", "sh"):
                out["ply_%s_%s" % (tag, f)] = np.asarray(gs[f])
        cams, imgs, pts = make_colmap(rng, 3, 7, 60)
        for name, raw in (("cameras", cams), ("images", imgs), ("points3D", pts)):
            open(os.path.join(tmp, name + ".bin"), "wb").write(raw)
            out["colmap_%s_bytes" % name] = np.frombuffer(raw, np.uint8)
        rc, ri = ref_rw.read_model(tmp, ext=".bin")                   # read_write_model.py:235-238
        ids = sorted(rc)
        out["cam_ids"] = np.array(ids)
        out["cam_models"] = np.array([rc[i].model for i in ids])
        out["cam_wh"] = np.array([[rc[i].width, rc[i].height] for i in ids])
        for i in ids:
            out["cam_params_%d" % i] = rc[i].params
        iids = list(ri)                                               # file order
        out["img_ids"] = np.array(iids)
        out["img_qvec"] = np.stack([ri[i].qvec for i in iids])
        out["img_tvec"] = np.stack([ri[i].tvec for i in iids])
        out["img_cam"] = np.array([ri[i].camera_id for i in iids])
        out["img_name"] = np.array([ri[i].name for i in iids])
        out["img_rotmat"] = np.stack([ri[i].qvec2rotmat() for i in iids])   # read_write_model.py:241-260
        out["img_npts"] = np.array([len(ri[i].point3D_ids) for i in iids])
        out["img_xys_cat"] = np.concatenate([ri[i].xys.reshape(-1, 2) for i in iids])
        out["img_p3d_cat"] = np.concatenate([ri[i].point3D_ids.reshape(-1) for i in iids])
        g = ref_rw.read_points_bin_as_gau(os.path.join(tmp, "points3D.bin"))  # read_write_model.py:179-232
        for f in ("pw", "rot", "scale", "alpha", "sh"):
            out["pts_%s" % f] = np.asarray(g[f])

    # pure-numpy helpers of gau_io.py
    R = []
    for k in range(64):
        q = rng.normal(0, 1, 4); q /= np.linalg.norm(q)
        R.append(ref_rw.qvec2rotmat(q))
    # rotations with trace near -1 exercise the three non-default branches (gau_io.py:33-57)
    for ax in range(3):
        for ang in (np.pi, np.pi - 1e-4, 3.0):
            a = np.zeros(3); a[ax] = 1
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            R.append(np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K)
    R = np.stack(R)
    out["m2q_R"] = R
    out["m2q_q"] = ref_io.matrix_to_quaternion(R)                     # gau_io.py:15-57
    fn_gs = ref_io.get_example_gs().copy()                            # gau_io.py:159-183
    out["example_gs_bytes"] = np.frombuffer(fn_gs.tobytes(), np.uint8)
    T = ref_rw.qvec2rotmat(np.array([0.5, -0.5, 0.5, 0.5]))
    rot_in = fn_gs.copy()
    rot_in["rot"] = np.array([[0.9, 0.1, 0.3, -0.2], [0.2, 0.7, -0.5, 0.4], [1, 0, 0, 0], [0.1, -0.6, 0.2, 0.75]],
                             np.float32)
    rot_in["rot"] /= np.linalg.norm(rot_in["rot"], axis=1, keepdims=True)
    out["rotate_T"] = T
    out["rotate_in_rot"] = rot_in["rot"].copy()
    got = ref_io.rotate_gaussian(T, rot_in.copy())                    # gau_io.py:108-127
    out["rotate_out_pw"] = got["pw"].copy()
    out["rotate_out_rot"] = got["rot"].copy()

    doc = ("G9: reference gau_io.load_ply / matrix_to_quaternion / rotate_gaussian / get_example_gs and "
           "read_write_model.read_model / read_points_bin_as_gau on synthesised files (bytes included).")
    _recipe.save("g9_io.npz", doc, **out)


if __name__ == "__main__":
    _recipe.begin("--check" in sys.argv[1:])
    main()
    sys.exit(_recipe.finish())
