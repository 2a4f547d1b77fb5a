"""IO / formats row: the PARSE layer pinned to the published file formats.

Fixture G9 (tests/golden/make_golden_io.py) had to drive the reference's readers through stand-ins for ``plyfile``
and ``faiss`` (absent from this image), so it pins what the reference does AFTER parsing, not the parsing.  Here the
files are written byte by byte from the format specifications themselves -- every field with its own
``struct.pack`` -- and the expected values are the Python literals that went in:

* PLY 1.0 (Turk, "The PLY Polygon File Format"): header of text lines ``ply`` / ``format <ascii|binary_little_endian|
  binary_big_endian> 1.0`` / ``comment`` / ``element <name> <count>`` / ``property <type> <name>`` /
  ``property list <count type> <item type> <name>`` / ``end_header``; elements in header order, properties in
  declaration order, scalar types char uchar short ushort int uint float double (and the int8 ... float64 aliases);
* the 3DGS exporter's vertex layout (x y z nx ny nz f_dc_0..2 f_rest_0..R-1 opacity scale_0..2 rot_0..3, f_rest
  CHANNEL-major) that reference gau_io.py:60-105 assumes and de-interleaves at gau_io.py:91;
* COLMAP's binary sparse model (doc "Output Format", src/colmap/scene/reconstruction_io.cc): cameras.bin /
  images.bin / points3D.bin as little-endian records, see ``easygaussiansplatting_amd/colmap.py``'s docstring.

The initial scale of ``read_points_bin_as_gau`` comes from faiss ``IndexFlatL2`` (read_write_model.py:216-222); its
published float32 arithmetic is restated in ``oracle/io_oracle.py::faiss_flat_l2_second`` and bounded here against
the exact metric (the GPU kernel is bounded against both in tests/test_gpu_io.py).
"""
import math
import os
import struct

import numpy as np
import pytest

from easygaussiansplatting_amd import colmap, gau_io
from oracle import io_oracle

SH_C0 = 0.28209479177387814


def _sigmoid(x):
    return 1.0 / (1.0 + math.exp(-x))


# ------------------------------------------------------------------------------------------------ PLY
def _gaussian_literals(i, n_rest):
    """Literal field values of vertex i (distinct everywhere: any mix-up of columns shows)."""
    v = {"x": 0.5 + i, "y": -1.25 * (i + 1), "z": 3.0 + 0.125 * i, "nx": 0.0, "ny": 0.0, "nz": 0.0,
         "opacity": -0.75 + 0.5 * i, "scale_0": -2.0 + 0.25 * i, "scale_1": -1.5, "scale_2": 0.125 * i,
         "rot_0": 2.0, "rot_1": 0.5 * i, "rot_2": -1.0, "rot_3": 0.25}
    for c in range(3):
        v["f_dc_%d" % c] = 0.1 * (c + 1) + i
    per = n_rest // 3
    for k in range(n_rest):
        ch, coef = divmod(k, per)                       # CHANNEL-major: all coefficients of red first
        v["f_rest_%d" % k] = 100.0 * ch + coef + 0.5 * i
    return v


def _expected(vals, n_rest):
    """What load_ply must return for those literals (activations of gau_io.py:78-84, de-interleave of :91)."""
    per = n_rest // 3
    rot = np.array([vals["rot_%d" % k] for k in range(4)])
    sh = [vals["f_dc_0"], vals["f_dc_1"], vals["f_dc_2"]]
    for coef in range(per):
        for ch in range(3):
            sh.append(vals["f_rest_%d" % (ch * per + coef)])     # record layout sh[3 * c + rgb]
    return dict(pw=[vals["x"], vals["y"], vals["z"]], rot=rot / np.linalg.norm(rot),
                scale=[math.exp(vals["scale_%d" % k]) for k in range(3)], alpha=_sigmoid(vals["opacity"]), sh=sh)


_STRUCT = {"float": "f", "double": "d", "uchar": "B", "int": "i", "short": "h", "float32": "f", "uint8": "B"}


def _ply_bytes(fmt, props, rows, eol="\n", extra_head=(), tail_elements=()):
    """A PLY file written field by field.  props: [(type, name)], rows: [dict name -> value]."""
    head = ["ply", "format %s 1.0" % fmt, "comment written field by field from the PLY 1.0 specification"]
    head += list(extra_head)
    head.append("element vertex %d" % len(rows))
    head += ["property %s %s" % (t, n) for t, n in props]
    for name, count, plist in tail_elements:
        head.append("element %s %d" % (name, count))
        head += plist
    head.append("end_header")
    out = (eol.join(head) + eol).encode("ascii")
    if fmt == "ascii":
        for r in rows:
            out += (" ".join(repr(r[n]) if t in ("float", "double", "float32") else str(int(r[n]))
                             for t, n in props) + "\n").encode("ascii")
        for name, count, plist in tail_elements:
            for k in range(count):
                out += b"3 0 1 2\n"
        return out
    order = "<" if fmt == "binary_little_endian" else ">"
    for r in rows:
        for t, n in props:
            out += struct.pack(order + _STRUCT[t], r[n])
    for name, count, plist in tail_elements:
        for k in range(count):
            out += struct.pack(order + "Biii", 3, 0, 1, 2)
    return out


def _props_3dgs(n_rest):
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % k for k in range(n_rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    return [("float", n) for n in names]


def _check(gs, rows, n_rest, rtol=2e-7):
    assert gs.shape == (len(rows),) and gs["sh"].shape == (len(rows), 3 + n_rest)
    for i, r in enumerate(rows):
        want = _expected(r, n_rest)
        for f in ("pw", "rot", "scale", "alpha", "sh"):
            np.testing.assert_allclose(np.asarray(gs[f][i], np.float64), np.asarray(want[f], np.float64), rtol=rtol,
                                       atol=1e-7, err_msg="vertex %d field %s" % (i, f))


@pytest.mark.parametrize("fmt", ["binary_little_endian", "binary_big_endian", "ascii"])
@pytest.mark.parametrize("n_rest", [45, 24, 9, 0])
def test_ply_3dgs_layout_from_the_specification(tmp_path, fmt, n_rest):
    rows = [_gaussian_literals(i, n_rest) for i in range(5)]
    fn = os.path.join(str(tmp_path), "a.ply")
    with open(fn, "wb") as f:
        f.write(_ply_bytes(fmt, _props_3dgs(n_rest), rows))
    _check(gau_io.load_ply(fn), rows, n_rest)


@pytest.mark.parametrize("fmt", ["binary_little_endian", "ascii"])
def test_ply_permuted_extra_and_typed_properties(tmp_path, fmt):
    """Properties are found BY NAME: any order, unrelated properties of other scalar types in between (their widths
    must be skipped correctly in the binary form), double-typed coordinates, other elements around the vertex
    element (one before it without list properties, a face element with a list property after it), CRLF header."""
    n_rest = 9
    rows = [_gaussian_literals(i, n_rest) for i in range(4)]
    for i, r in enumerate(rows):
        r.update(red=200 + i, flag=-3 + i, confidence=0.5 * i, label=7 - i)
    base = [n for _, n in _props_3dgs(n_rest)]
    order = base[::-1]                                   # reversed: rot_3 first, x last
    props = []
    for k, n in enumerate(order):
        props.append(("double" if n in ("x", "y", "z") else "float", n))
        if k == 2: props.append(("uchar", "red"))
        if k == 5: props.append(("short", "flag"))
        if k == 11: props.append(("double", "confidence"))
        if k == 20: props.append(("int", "label"))
    order_c = "<"
    raw = _ply_bytes(fmt, props, rows, eol="\r\n",
                     tail_elements=[("face", 2, ["property list uchar int vertex_indices"])])
    # an element BEFORE the vertex element (fixed-size properties): its data precedes the vertex data
    marker = b"element vertex"
    head_extra = b"element calib 2\r\nproperty double k1\r\nproperty uchar ok\r\n"
    raw = raw.replace(marker, head_extra + marker, 1)
    body_at = raw.index(b"end_header\r\n") + len(b"end_header\r\n")
    calib = (b"0.25 1\n-0.5 0\n" if fmt == "ascii" else
             struct.pack(order_c + "dB", 0.25, 1) + struct.pack(order_c + "dB", -0.5, 0))
    raw = raw[:body_at] + calib + raw[body_at:]
    fn = os.path.join(str(tmp_path), "p.ply")
    with open(fn, "wb") as f:
        f.write(raw)
    _check(gau_io.load_ply(fn), rows, n_rest)
    v = gau_io.read_ply_vertices(fn)
    assert [int(x) for x in v["red"]] == [200, 201, 202, 203] and [int(x) for x in v["flag"]] == [-3, -2, -1, 0]
    assert v["x"].dtype.itemsize == 8 or fmt == "ascii"


def test_ply_malformed_inputs_raise(tmp_path):
    fn = os.path.join(str(tmp_path), "bad.ply")
    rows = [_gaussian_literals(0, 0)]
    good = _ply_bytes("binary_little_endian", _props_3dgs(0), rows)
    for label, data in (("no magic", b"plx" + good[3:]), ("truncated", good[:-5]),
                        ("no end_header", good.replace(b"end_header", b"end_headxx")),
                        ("missing property", good.replace(b"property float opacity\n", b"")),
                        ("unknown type", good.replace(b"property float opacity", b"property half opacity")),
                        ("f_rest not a multiple of 3", _ply_bytes("binary_little_endian", _props_3dgs(0)[:9] +
                                                                  [("float", "f_rest_0")] + _props_3dgs(0)[9:],
                                                                  [dict(rows[0], f_rest_0=1.0)]))):
        with open(fn, "wb") as f:
            f.write(data)
        with pytest.raises(gau_io.PlyFormatError):
            gau_io.load_ply(fn)


# ------------------------------------------------------------------------------------------------ COLMAP
CAMS = [  # (id, model id, width, height, params)       model ids: src/colmap/sensor/models.h
    (1, 1, 1957, 1091, (1163.25, 1156.5, 978.5, 545.5)),            # PINHOLE fx fy cx cy
    (7, 0, 640, 480, (525.0, 319.5, 239.5)),                        # SIMPLE_PINHOLE f cx cy
    (3, 2, 800, 600, (700.0, 400.0, 300.0, -0.0625)),               # SIMPLE_RADIAL f cx cy k
    (12, 4, 1280, 720, (900.0, 905.0, 640.0, 360.0, 0.01, -0.02, 0.003, -0.004)),   # OPENCV
]
IMAGES = [  # (id, qvec wxyz, tvec, camera id, name, [(x, y, point3D id)])
    (4, (0.5, -0.5, 0.5, 0.5), (1.5, -2.25, 3.125), 1, "frames/000004.jpg", [(10.5, 20.25, 17), (99.0, 0.5, -1)]),
    (2, (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 7, "b.png", []),
    (9, (0.0, 0.6, 0.0, 0.8), (-7.0, 8.5, 0.03125), 12, "üñi.jpg", [(1.0, 2.0, 3), (4.0, 5.0, 6), (7.0, 8.0, 9)]),
]
POINTS = [  # (id, xyz, rgb, error, [(image id, point2D idx)])
    (17, (0.25, -1.5, 2.0), (255, 0, 128), 0.75, [(4, 0), (9, 2)]),
    (3, (10.0, 20.0, -30.0), (1, 2, 3), 1.5, []),
    (6, (-0.125, 0.0625, 4.5), (17, 200, 99), 0.03125, [(9, 1)]),
    (2 ** 40 + 5, (1e-3, 2e3, -4.0), (0, 255, 255), 2.0, [(2, 0), (4, 1), (9, 0)]),
]


def _cameras_bin():
    out = struct.pack("<Q", len(CAMS))
    for cid, model, w, h, params in CAMS:
        out += struct.pack("<i", cid) + struct.pack("<i", model) + struct.pack("<Q", w) + struct.pack("<Q", h)
        for p in params:
            out += struct.pack("<d", p)
    return out


def _images_bin():
    out = struct.pack("<Q", len(IMAGES))
    for iid, q, t, cam, name, pts in IMAGES:
        out += struct.pack("<i", iid)
        for v in q + t:
            out += struct.pack("<d", v)
        out += struct.pack("<i", cam) + name.encode("utf-8") + b"\x00" + struct.pack("<Q", len(pts))
        for x, y, pid in pts:
            out += struct.pack("<d", x) + struct.pack("<d", y) + struct.pack("<q", pid)
    return out


def _points_bin():
    out = struct.pack("<Q", len(POINTS))
    for pid, xyz, rgb, err, track in POINTS:
        out += struct.pack("<Q", pid)
        for v in xyz:
            out += struct.pack("<d", v)
        for c in rgb:
            out += struct.pack("<B", c)
        out += struct.pack("<d", err) + struct.pack("<Q", len(track))
        for im, idx in track:
            out += struct.pack("<i", im) + struct.pack("<i", idx)
    return out


def _model_dir(tmp_path):
    d = str(tmp_path)
    for name, data in (("cameras.bin", _cameras_bin()), ("images.bin", _images_bin()), ("points3D.bin", _points_bin())):
        with open(os.path.join(d, name), "wb") as f:
            f.write(data)
    return d


def test_colmap_binary_model_from_the_specification(tmp_path):
    d = _model_dir(tmp_path)
    cams, imgs = colmap.read_model(d)
    assert list(cams) == [c[0] for c in CAMS] and list(imgs) == [i[0] for i in IMAGES]       # file order kept
    names = {0: "SIMPLE_PINHOLE", 1: "PINHOLE", 2: "SIMPLE_RADIAL", 4: "OPENCV"}
    for cid, model, w, h, params in CAMS:
        c = cams[cid]
        assert (c.id, c.model, c.width, c.height) == (cid, names[model], w, h)
        assert c.params.dtype == np.float64 and c.params.tolist() == list(params)
    for iid, q, t, cam, name, pts in IMAGES:
        im = imgs[iid]
        assert im.id == iid and im.camera_id == cam and im.name == name
        assert im.qvec.tolist() == list(q) and im.tvec.tolist() == list(t)
        assert im.xys.shape == (len(pts), 2) and im.point3D_ids.dtype == np.int64
        assert im.xys.tolist() == [[x, y] for x, y, _ in pts] and im.point3D_ids.tolist() == [p for _, _, p in pts]
    # qvec2rotmat: (0.5, -0.5, 0.5, 0.5) is the rotation x -> -z, y -> x ... written out by hand
    R = imgs[4].qvec2rotmat()
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-15)
    np.testing.assert_allclose(R, [[0.0, -1.0, 0.0], [0.0, 0.0, 1.0], [-1.0, 0.0, 0.0]], atol=1e-15)
    ids, xyz, rgb, err, track = colmap.read_points3D_binary(os.path.join(d, "points3D.bin"))
    assert ids.tolist() == [p[0] for p in POINTS] and ids.dtype == np.uint64
    assert xyz.tolist() == [list(p[1]) for p in POINTS] and rgb.tolist() == [list(p[2]) for p in POINTS]
    assert err.tolist() == [p[3] for p in POINTS] and track.tolist() == [len(p[4]) for p in POINTS]
    # the initial Gaussians (read_write_model.py:197-232) from those literals
    gs = colmap.read_points_bin_as_gau(os.path.join(d, "points3D.bin"), nn_sqdist=io_oracle.nn_sqdist)
    assert gs["pw"].tolist() == [[np.float32(v) for v in p[1]] for p in POINTS]
    np.testing.assert_allclose(gs["sh"], [[(c / 255 - 0.5) / SH_C0 for c in p[2]] for p in POINTS], rtol=1e-6)
    assert gs["alpha"].tolist() == [np.float32(0.8)] * 4 and gs["rot"].tolist() == [[1, 0, 0, 0]] * 4
    assert (gs["scale"] == 3.0).all()                      # nearest neighbours are > sqrt(3) apart: clipped at 3


def test_colmap_truncated_and_unknown_model_raise(tmp_path):
    d = _model_dir(tmp_path)
    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        fn = os.path.join(d, name)
        data = open(fn, "rb").read()
        with open(fn, "wb") as f:
            f.write(data[:-3])
        with pytest.raises(colmap.ColmapFormatError):
            {"cameras.bin": colmap.read_cameras_binary, "images.bin": colmap.read_images_binary,
             "points3D.bin": colmap.read_points3D_binary}[name](fn)
    fn = os.path.join(d, "cameras.bin")
    with open(fn, "wb") as f:
        f.write(struct.pack("<Q", 1) + struct.pack("<iiQQ", 1, 99, 10, 10))
    with pytest.raises(colmap.ColmapFormatError):
        colmap.read_cameras_binary(fn)


# ------------------------------------------------------------------------------------------------ faiss IndexFlatL2
def test_faiss_flat_l2_formula_against_the_exact_metric():
    """faiss.IndexFlatL2.search computes, for more than 20 queries, ||x||^2 + ||y||^2 - 2 <x, y> in float32 (norms by
    fvec_norms_L2sqr, inner products by sgemm, negative results clamped to 0: faiss/utils/distances.cpp,
    exhaustive_L2sqr_blas) and read_write_model.py:219-222 takes the SECOND smallest value per row (the first is the
    query itself).  Its error is bounded by the float32 rounding of the NORMS, not of the distance: a cloud whose
    points sit R units from the origin gets initial scales (the value clipped to [0.01, 3]) within ~2e-6 R^2 of the
    exact metric -- 1e-3 at R = 25, which is why the HIP kernel (differences first, then squares) is compared with
    the exact metric and only BOUNDED against this formula."""
    rng = np.random.default_rng(3)
    for n, spread, offset in ((400, 1.0, 0.0), (1500, 8.0, 0.0), (1500, 3.0, 25.0)):
        p = (rng.standard_normal((n, 3)) * spread + offset).astype(np.float32)
        exact = io_oracle.nn_sqdist(p)
        fa = io_oracle.faiss_flat_l2_second(p)
        assert fa.dtype == np.float32 and (fa >= 0).all()
        norms = (p.astype(np.float64) ** 2).sum(1)
        bound = 16 * np.finfo(np.float32).eps * (norms + norms.max())          # a few roundings of values of that size
        assert (np.abs(fa - exact) <= bound + 1e-12).all(), np.abs(fa - exact).max()
        a, b = np.clip(fa, 0.01, 3), np.clip(exact, 0.01, 3)      # what read_points_bin_as_gau keeps of it
        assert (np.abs(a - b) <= bound + 1e-12).all() and np.abs(a - b).max() <= 2e-6 * norms.max() + 1e-6
    # a duplicated point: the second "neighbour" of both copies is the other copy, at distance ~0
    p = np.array([[1, 2, 3], [1, 2, 3], [4, 5, 6], [0, 0, 1]] + [[10 + k, 0, 0] for k in range(20)], np.float32)
    assert io_oracle.faiss_flat_l2_second(p)[:2].tolist() == [0.0, 0.0] and io_oracle.nn_sqdist(p)[:2].tolist() == [0.0, 0.0]
