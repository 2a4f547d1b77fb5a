"""IO / formats row: the repo's readers against fixture G9 = outputs of the
reference's gau_io.py / read_write_model.py on the byte strings stored in the fixture; plus
round trips and malformed-input behaviour.  Host-only (no GPU)."""
import os

import numpy as np
import pytest

from easygaussiansplatting_amd import colmap, gau_io
from easygaussiansplatting_amd.scene import gsdata_type
from oracle import io_oracle
from tests.conftest import load_golden


def _write(tmp_path, name, arr):
    fn = os.path.join(str(tmp_path), name)
    with open(fn, "wb") as f:
        f.write(arr.tobytes())
    return fn


@pytest.mark.parametrize("tag,sh", [("deg3", 48), ("deg2", 27), ("deg1", 12)])
def test_load_ply_matches_reference(tmp_path, tag, sh):
    g = load_golden("g9_io.npz")
    gs = gau_io.load_ply(_write(tmp_path, tag + ".ply", g["ply_%s_bytes" % tag]))
    assert gs.dtype == np.dtype(gsdata_type(sh))
    for f in ("pw", "rot", "scale", "alpha", "sh"):
        np.testing.assert_allclose(gs[f], g["ply_%s_%s" % (tag, f)], rtol=1e-6, atol=1e-7, err_msg=f)
    assert gau_io.load_gs(os.path.join(str(tmp_path), tag + ".ply")).shape == gs.shape


def test_ply_round_trip_and_other_encodings(tmp_path):
    g = load_golden("g9_io.npz")
    gs = gau_io.load_ply(_write(tmp_path, "a.ply", g["ply_deg3_bytes"]))
    out = os.path.join(str(tmp_path), "b.ply")
    gau_io.save_ply(out, gs)
    back = gau_io.load_ply(out)
    for f in ("pw", "rot", "scale", "alpha", "sh"):
        np.testing.assert_allclose(back[f], gs[f], rtol=2e-6, atol=1e-6, err_msg=f)
    # degree 0 (the reference raises on such a file, gau_io.py:91) and an ascii / big-endian file
    gs0 = gau_io.get_example_gs()
    gau_io.save_ply(out, gs0)
    b0 = gau_io.load_ply(out)
    assert b0["sh"].shape == (4, 3)
    np.testing.assert_allclose(b0["sh"], gs0["sh"], rtol=1e-6)
    v = gau_io.read_ply_vertices(out)
    names = v.dtype.names
    hdr = "ply\nformat ascii 1.0\ncomment made by a test\nelement vertex %d\n" % len(v)
    hdr += "".join("property float %s\n" % k for k in names) + "element face 0\nproperty list uchar int vertex_indices\nend_header\n"
    body = "\n".join(" ".join(repr(float(x)) for x in row) for row in v.tolist()) + "\n"
    with open(out, "w") as f:
        f.write(hdr + body)
    np.testing.assert_allclose(gau_io.load_ply(out)["scale"], b0["scale"], rtol=1e-6)
    be = np.array(v.tolist(), dtype=">f4")
    with open(out, "wb") as f:
        f.write(("ply\nformat binary_big_endian 1.0\nelement vertex %d\n" % len(v)).encode())
        f.write("".join("property float %s\n" % k for k in names).encode() + b"end_header\n" + be.tobytes())
    np.testing.assert_allclose(gau_io.load_ply(out)["rot"], b0["rot"], rtol=1e-6)


def test_ply_malformed(tmp_path):
    g = load_golden("g9_io.npz")
    raw = g["ply_deg1_bytes"].tobytes()
    with pytest.raises(gau_io.PlyFormatError):
        gau_io.load_ply(_write(tmp_path, "t.ply", np.frombuffer(raw[:-10], np.uint8)))       # truncated
    with pytest.raises(gau_io.PlyFormatError):
        gau_io.load_ply(_write(tmp_path, "n.ply", np.frombuffer(b"plx" + raw[3:], np.uint8)))
    with pytest.raises(gau_io.PlyFormatError):
        gau_io.load_ply(_write(tmp_path, "m.ply", np.frombuffer(raw.replace(b"opacity", b"opacitx"), np.uint8)))
    with pytest.raises(ValueError):
        gau_io.load_gs("scene.obj")


def test_npy_record_round_trip(tmp_path):
    g = load_golden("g9_io.npz")
    ex = gau_io.get_example_gs()
    assert ex.tobytes() == g["example_gs_bytes"].tobytes() and ex.dtype == np.dtype(gsdata_type(3))
    fn = os.path.join(str(tmp_path), "gs.npy")
    gau_io.save_gs(fn, ex)
    back = gau_io.load_gs(fn)
    assert back.dtype == ex.dtype and back.tobytes() == ex.tobytes()


def test_matrix_to_quaternion_and_rotate_gaussian():
    g = load_golden("g9_io.npz")
    q = gau_io.matrix_to_quaternion(g["m2q_R"])
    np.testing.assert_allclose(q, g["m2q_q"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gau_io.quaternion_to_matrix(q), g["m2q_R"], atol=1e-9)
    gs = gau_io.get_example_gs().copy()
    gs["rot"] = g["rotate_in_rot"]
    out = gau_io.rotate_gaussian(g["rotate_T"], gs)
    np.testing.assert_allclose(out["pw"], g["rotate_out_pw"], atol=1e-7)
    np.testing.assert_allclose(out["rot"], g["rotate_out_rot"], atol=1e-6)


def test_colmap_readers_match_reference(tmp_path):
    g = load_golden("g9_io.npz")
    for name in ("cameras", "images", "points3D"):
        _write(tmp_path, name + ".bin", g["colmap_%s_bytes" % name])
    cams, imgs = colmap.read_model(str(tmp_path))
    assert sorted(cams) == list(g["cam_ids"])
    for k, i in enumerate(sorted(cams)):
        assert cams[i].model == str(g["cam_models"][k]) and cams[i].id == i
        assert [cams[i].width, cams[i].height] == list(g["cam_wh"][k])
        np.testing.assert_array_equal(cams[i].params, g["cam_params_%d" % i])
    assert list(imgs) == list(g["img_ids"])                       # file order
    vals = list(imgs.values())
    np.testing.assert_array_equal(np.stack([v.qvec for v in vals]), g["img_qvec"])
    np.testing.assert_array_equal(np.stack([v.tvec for v in vals]), g["img_tvec"])
    assert [v.camera_id for v in vals] == list(g["img_cam"]) and [v.name for v in vals] == list(g["img_name"])
    np.testing.assert_allclose(np.stack([v.qvec2rotmat() for v in vals]), g["img_rotmat"], rtol=0, atol=1e-15)
    assert [len(v.point3D_ids) for v in vals] == list(g["img_npts"])
    np.testing.assert_array_equal(np.concatenate([v.xys.reshape(-1, 2) for v in vals]), g["img_xys_cat"])
    np.testing.assert_array_equal(np.concatenate([v.point3D_ids for v in vals]), g["img_p3d_cat"])
    # initial Gaussians; the neighbour search is injected (the HIP kernel is covered by the GPU test)
    gs = colmap.read_points_bin_as_gau(os.path.join(str(tmp_path), "points3D.bin"), nn_sqdist=io_oracle.nn_sqdist)
    assert gs.dtype == np.dtype(gsdata_type(3))
    for f in ("pw", "rot", "alpha", "sh"):
        np.testing.assert_array_equal(gs[f], g["pts_" + f], err_msg=f)
    np.testing.assert_allclose(gs["scale"], g["pts_scale"], rtol=1e-6)
    assert gs["scale"].min() == np.float32(0.01) and gs["scale"].max() == np.float32(3.0)   # both clips bind


def test_colmap_malformed(tmp_path):
    g = load_golden("g9_io.npz")
    with pytest.raises(colmap.ColmapFormatError):
        colmap.read_images_binary(_write(tmp_path, "i.bin", g["colmap_images_bytes"][:-5]))
    with pytest.raises(colmap.ColmapFormatError):
        colmap.read_cameras_binary(_write(tmp_path, "c.bin", g["colmap_cameras_bytes"][:30]))
    bad = g["colmap_cameras_bytes"].copy()
    bad[12] = 99                                                  # model id of the first camera
    with pytest.raises(colmap.ColmapFormatError):
        colmap.read_cameras_binary(_write(tmp_path, "m.bin", bad))
    empty = np.zeros(8, np.uint8)
    assert colmap.read_cameras_binary(_write(tmp_path, "e.bin", empty)) == {}
    assert colmap.read_points_bin_as_gau(_write(tmp_path, "p.bin", empty), nn_sqdist=io_oracle.nn_sqdist).shape == (0,)


def test_nn_oracle_against_kdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(1)
    for n in (2, 100, 5000):
        p = rng.normal(0, 1, (n, 3)).astype(np.float32)
        d, _ = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=2)
        np.testing.assert_allclose(io_oracle.nn_sqdist(p), d[:, 1] ** 2, rtol=1e-6, atol=1e-9)
