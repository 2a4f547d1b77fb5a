"""Gaussian-set file formats: the counterpart of the reference's ``gsplat/gau_io.py``.

* structured-record ``.npy`` -- ``load_gs`` / ``save_gs`` / ``save_training_params``
  (gau_io.py:130-156), record dtype ``gsdata_type`` (gau_io.py:7-12);
* 3DGS ``point_cloud.ply`` -- ``load_ply`` (gau_io.py:60-105) with its own PLY reader (the
  reference depends on the ``plyfile`` package) and, new, ``save_ply`` (the inverse);
* ``matrix_to_quaternion`` (gau_io.py:15-57), ``rotate_gaussian`` (gau_io.py:108-127),
  ``get_example_gs`` (gau_io.py:159-183).

A ``.ply`` stores UN-activated values: ``opacity`` = logit(alpha), ``scale_*`` = log(scale),
``rot_*`` unnormalised (w, x, y, z), SH as ``f_dc_{0..2}`` (degree 0, rgb) and
``f_rest_{0..R-1}`` CHANNEL-major (all R/3 red coefficients, then green, then blue); the record
keeps activated values and SH interleaved ``sh[3 * c + rgb]`` (gau_io.py:91).
"""
from __future__ import annotations

import numpy as np

from .scene import example_gs, gsdata_type

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


class PlyFormatError(ValueError):
    pass


def read_ply_vertices(path) -> np.ndarray:
    """Structured array of the ``vertex`` element of a PLY file (ascii, binary little or big endian).
    Elements that precede ``vertex`` must not contain list properties (their size is then unknown
    without parsing them); elements after it are ignored."""
    with open(path, "rb") as f:
        raw = f.read()
    if not raw.startswith(b"ply"):
        raise PlyFormatError("%s: not a PLY file" % path)
    marker = raw.find(b"end_header")
    if marker < 0:
        raise PlyFormatError("%s: no end_header" % path)
    data_at = raw.index(b"\n", marker) + 1
    fmt = None
    elements = []                       # [name, count, [(prop, dtype)], has_list]
    for ln in raw[:marker].decode("ascii", "replace").splitlines()[1:]:
        t = ln.split()
        if not t or t[0] in ("comment", "obj_info"):
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elements.append([t[1], int(t[2]), [], False])
        elif t[0] == "property":
            if not elements:
                raise PlyFormatError("%s: property before element" % path)
            if t[1] == "list":
                elements[-1][3] = True
            else:
                if t[1] not in _PLY_TYPES:
                    raise PlyFormatError("%s: unknown property type %s" % (path, t[1]))
                elements[-1][2].append((t[2], _PLY_TYPES[t[1]]))
    if fmt not in ("binary_little_endian", "binary_big_endian", "ascii"):
        raise PlyFormatError("%s: unsupported format %r" % (path, fmt))
    order = {"binary_little_endian": "<", "binary_big_endian": ">", "ascii": "<"}[fmt]
    offset = data_at
    tokens = raw[data_at:].split() if fmt == "ascii" else None
    tok_at = 0
    for name, count, props, has_list in elements:
        if has_list and name == "vertex":
            raise PlyFormatError("%s: list property in the vertex element" % path)
        dtype = np.dtype([(p, order + ty) for p, ty in props])
        if name == "vertex":
            if fmt == "ascii":
                need = count * len(props)
                vals = tokens[tok_at:tok_at + need]
                if len(vals) != need:
                    raise PlyFormatError("%s: truncated vertex data" % path)
                cols = np.array(vals, dtype="f8").reshape(count, len(props))
                out = np.empty(count, dtype=dtype)
                for k, (p, _) in enumerate(props):
                    out[p] = cols[:, k]
                return out
            if offset + count * dtype.itemsize > len(raw):
                raise PlyFormatError("%s: truncated vertex data" % path)
            return np.frombuffer(raw, dtype=dtype, count=count, offset=offset)
        if has_list:
            raise PlyFormatError("%s: list element %r precedes the vertex element" % (path, name))
        offset += count * dtype.itemsize
        tok_at += count * len(props)
    raise PlyFormatError("%s: no vertex element" % path)


def load_ply(path, T=None) -> np.ndarray:
    """3DGS ``.ply`` -> record array of activated Gaussians (gau_io.py:60-105).

    The SH width is the number of ``f_dc_*`` + ``f_rest_*`` properties (the reference computes
    ``len(row) - 14``, the same number for files that carry ``nx ny nz``); a file without
    ``f_rest_*`` (degree 0), on which the reference raises at gau_io.py:91, loads with ``sh`` [N,3]."""
    v = read_ply_vertices(path)
    names = v.dtype.names
    for k in ("x", "y", "z", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3",
              "f_dc_0", "f_dc_1", "f_dc_2"):
        if k not in names:
            raise PlyFormatError("%s: missing vertex property %s" % (path, k))
    n_rest = sum(1 for k in names if k.startswith("f_rest_"))
    if n_rest % 3:
        raise PlyFormatError("%s: %d f_rest properties (not a multiple of 3)" % (path, n_rest))
    col = lambda *ks: np.stack([np.asarray(v[k]) for k in ks], axis=1)
    pws = col("x", "y", "z")
    alphas = 1 / (1 + np.exp(-np.asarray(v["opacity"])))
    scales = np.exp(col("scale_0", "scale_1", "scale_2"))
    rots = col("rot_0", "rot_1", "rot_2", "rot_3")
    rots = rots / np.linalg.norm(rots, axis=1)[:, np.newaxis]
    shs = np.zeros([pws.shape[0], 3 + n_rest])
    shs[:, :3] = col("f_dc_0", "f_dc_1", "f_dc_2")
    if n_rest:
        rest = col(*["f_rest_%d" % i for i in range(n_rest)])
        # channel-major [rgb][coef] -> interleaved [coef][rgb]
        shs[:, 3:] = rest.reshape(-1, 3, n_rest // 3).transpose(0, 2, 1).reshape(-1, n_rest)
    f = np.float32
    gs = np.rec.fromarrays([pws.astype(f), rots.astype(f), scales.astype(f), alphas.astype(f), shs.astype(f)],
                           dtype=gsdata_type(3 + n_rest))
    return gs


def save_ply(path, gs) -> None:
    """Inverse of ``load_ply``: write the record array as a binary little-endian 3DGS ``.ply``
    (with zero normals, as the 3DGS exporter does).  Not in the reference (it only reads)."""
    n = gs.shape[0]
    shs = np.asarray(gs["sh"], np.float64).reshape(n, -1)
    n_rest = shs.shape[1] - 3
    if n_rest < 0 or n_rest % 3:
        raise ValueError("sh width %d is not 3 * (degree + 1)^2" % shs.shape[1])
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += ["f_rest_%d" % i for i in range(n_rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    rows = np.zeros((n, len(names)), "<f4")
    rows[:, 0:3] = gs["pw"]
    rows[:, 6:9] = shs[:, :3]
    if n_rest:
        rows[:, 9:9 + n_rest] = shs[:, 3:].reshape(n, n_rest // 3, 3).transpose(0, 2, 1).reshape(n, n_rest)
    a = np.clip(np.asarray(gs["alpha"], np.float64).reshape(n), 1e-7, 1 - 1e-7)
    rows[:, 9 + n_rest] = np.log(a / (1 - a))
    rows[:, 10 + n_rest:13 + n_rest] = np.log(np.asarray(gs["scale"], np.float64))
    rows[:, 13 + n_rest:17 + n_rest] = gs["rot"]
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    hdr += "".join("property float %s\n" % k for k in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(rows.tobytes())


def matrix_to_quaternion(matrices) -> np.ndarray:
    """Rotation matrices [N,3,3] -> quaternions [N,4] (w, x, y, z), branch per row on the trace /
    largest diagonal element exactly as gau_io.py:15-57 selects them."""
    m = np.asarray(matrices, dtype=np.float64)
    d0, d1, d2 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    t = 1 + d0 + d1 + d2
    q = np.ones((m.shape[0], 4), np.float64)
    by_trace = t > 0.0000001
    x_big = ~by_trace & (d0 > d1) & (d0 > d2)
    y_big = ~by_trace & ~((d0 > d1) & (d0 > d2)) & (d1 > d2)
    z_big = ~by_trace & ~x_big & ~y_big
    # antisymmetric / symmetric off-diagonal combinations
    ax, ay, az = m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1]
    sxy, sxz, syz = m[:, 0, 1] + m[:, 1, 0], m[:, 0, 2] + m[:, 2, 0], m[:, 1, 2] + m[:, 2, 1]
    with np.errstate(invalid="ignore", divide="ignore"):
        s = 0.5 / np.sqrt(t)
        q[by_trace] = np.stack([0.25 / s, ax * s, ay * s, az * s], 1)[by_trace]
        s = 2.0 * np.sqrt(1.0 + d0 - d1 - d2)
        q[x_big] = np.stack([ax / s, 0.25 * s, sxy / s, sxz / s], 1)[x_big]
        s = 2.0 * np.sqrt(1.0 + d1 - d0 - d2)
        q[y_big] = np.stack([ay / s, sxy / s, 0.25 * s, syz / s], 1)[y_big]
        s = 2.0 * np.sqrt(1.0 + d2 - d0 - d1)
        q[z_big] = np.stack([az / s, sxz / s, syz / s, 0.25 * s], 1)[z_big]
    return q.astype(np.asarray(matrices).dtype if np.asarray(matrices).dtype.kind == "f" else np.float64)


def quaternion_to_matrix(rots) -> np.ndarray:
    """(w, x, y, z) [N,4] -> [N,3,3]; no normalisation (the matrix of gau_io.py:115-119)."""
    w, x, y, z = (np.asarray(rots)[:, i] for i in range(4))
    return np.array([
        [1.0 - 2 * (y ** 2 + z ** 2), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1.0 - 2 * (x ** 2 + z ** 2), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1.0 - 2 * (x ** 2 + y ** 2)]]).transpose(2, 0, 1)


def rotate_gaussian(T, gs):
    """Apply the 3x3 transform ``T`` to positions and orientations in place (gau_io.py:108-127)."""
    T = np.asarray(T)
    gs["pw"] = (T @ gs["pw"].T).T
    gs["rot"] = matrix_to_quaternion(T @ quaternion_to_matrix(gs["rot"]))
    return gs


def load_gs(fn) -> np.ndarray:
    """gau_io.py:130-137; an unsupported extension raises instead of exiting the interpreter."""
    fn = str(fn)
    if fn.endswith(".ply"):
        return load_ply(fn)
    if fn.endswith(".npy"):
        return np.load(fn)
    raise ValueError("%s is not a supported file." % fn)


def save_gs(fn, gs) -> None:
    """gau_io.py:140-141."""
    np.save(fn, gs)


def save_training_params(fn, training_params) -> np.ndarray:
    """Checkpoint of ACTIVATED parameters (gau_io.py:141-156); returns the record array."""
    import torch
    with torch.no_grad():
        p = training_params
        shs = torch.cat((p["low_shs"], p["high_shs"]), dim=1).detach().cpu().numpy()
        arrs = [p["pws"].detach().cpu().numpy(),
                torch.nn.functional.normalize(p["rots_raw"]).detach().cpu().numpy(),
                torch.exp(p["scales_raw"]).detach().cpu().numpy(),
                torch.sigmoid(p["alphas_raw"]).detach().cpu().numpy().reshape(-1), shs]
    gs = np.rec.fromarrays(arrs, dtype=gsdata_type(shs.shape[1]))
    np.save(fn, gs)
    return gs


def get_example_gs() -> np.ndarray:
    """The four-Gaussian example of gau_io.py:159-183 as a record array."""
    sc = example_gs()
    return np.rec.fromarrays([sc.pws, sc.rots, sc.scales, sc.alphas, sc.shs], dtype=gsdata_type(sc.shs.shape[1]))
