"""GPU side of the IO / formats row: the nearest-neighbour kernel against the
float64 oracle and the COLMAP dataset -> trainer path (train.py counterpart)."""
import os

import numpy as np
import pytest
import torch

from oracle import io_oracle
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 255, 1024, 1025, 5000, 40000])
def test_nn_sqdist_matches_oracle(n):
    from easygaussiansplatting_amd.knn import nn_sqdist
    rng = np.random.default_rng(n)
    p = (rng.normal(0, 1, (n, 3)) * rng.uniform(0.1, 10)).astype(np.float32)
    if n > 10:
        p[5] = p[9]                                   # duplicate -> 0
    got = nn_sqdist(p).cpu().numpy()
    if n == 1:
        assert got[0] > 1e37                          # no other point
        return
    want = io_oracle.nn_sqdist(p)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7)
    if n > 10:
        assert got[5] == 0 and got[9] == 0


def test_nn_sqdist_within_the_bound_of_the_faiss_formula():
    """read_write_model.py:216-222 reads column 1 of ``faiss.IndexFlatL2.search(pws, 2)``.  faiss evaluates
    ||x||^2 + ||y||^2 - 2 <x, y> in float32 (restated in oracle/io_oracle.py::faiss_flat_l2_second), whose rounding
    error scales with the NORMS; the kernel subtracts first.  Both sit inside that bound of the exact metric, and the
    initial scales (the value clipped to [0.01, 3]) they lead to differ by less than the bound too."""
    from easygaussiansplatting_amd.knn import nn_sqdist
    rng = np.random.default_rng(11)
    for n, spread, offset in ((3000, 2.0, 0.0), (3000, 3.0, 20.0)):
        p = (rng.standard_normal((n, 3)) * spread + offset).astype(np.float32)
        got = nn_sqdist(p).cpu().numpy().astype(np.float64)
        exact = io_oracle.nn_sqdist(p)
        fa = io_oracle.faiss_flat_l2_second(p).astype(np.float64)
        norms = (p.astype(np.float64) ** 2).sum(1)
        bound = 16 * np.finfo(np.float32).eps * (norms + norms.max())
        assert (np.abs(got - exact) <= 2e-5 * exact + 1e-7).all()              # the kernel: relative to the DISTANCE
        assert (np.abs(fa - exact) <= bound).all() and (np.abs(fa - got) <= bound + 2e-5 * exact).all()
        assert np.abs(np.clip(fa, 0.01, 3) - np.clip(got, 0.01, 3)).max() <= bound.max()


def test_points_to_gaussians_on_device_matches_reference(tmp_path):
    from easygaussiansplatting_amd import colmap
    g = load_golden("g9_io.npz")
    fn = os.path.join(str(tmp_path), "points3D.bin")
    open(fn, "wb").write(g["colmap_points3D_bytes"].tobytes())
    gs = colmap.read_points_bin_as_gau(fn)            # HIP neighbour search
    for f in ("pw", "rot", "alpha", "sh"):
        np.testing.assert_array_equal(gs[f], g["pts_" + f])
    np.testing.assert_allclose(gs["scale"], g["pts_scale"], rtol=2e-5)


def test_nn_sqdist_large_cloud_sampled():
    """COLMAP-sized cloud (200 k points): sampled rows against a KD-tree."""
    from scipy.spatial import cKDTree
    from easygaussiansplatting_amd.knn import nn_sqdist
    rng = np.random.default_rng(0)
    p = rng.normal(0, 3, (200_000, 3)).astype(np.float32)
    dev = torch.from_numpy(p).cuda()
    nn_sqdist(dev)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); got = nn_sqdist(dev); t1.record(); torch.cuda.synchronize()
    print("nn_sqdist 200k points: %.2f ms" % t0.elapsed_time(t1))
    idx = rng.choice(len(p), 4000, replace=False)
    d, _ = cKDTree(p.astype(np.float64)).query(p[idx].astype(np.float64), k=2)
    np.testing.assert_allclose(got.cpu().numpy()[idx], d[:, 1] ** 2, rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize("model", ["PINHOLE", "SIMPLE_PINHOLE"])
def test_dataset_from_colmap_scene_and_training(tmp_path, model):
    from easygaussiansplatting_amd import gsplatcu as gsc
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.dataset import GSplatDataset
    from easygaussiansplatting_amd.function import Camera, render
    from easygaussiansplatting_amd.trainer import Trainer
    from tests.colmap_fixture import write_scene
    gsc.set_policy("gsplatcu")
    sc = S.small_scene(2000, 96, 64, 3, seed=4)
    cams = S.ring_cameras(sc.cam, 4, radius=5.0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    imgs = []
    with torch.no_grad():
        for c in cams:
            im = render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), Camera.from_scene(c))[0]
            imgs.append((im.clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255 + 0.5).astype(np.uint8))
    rgb = np.clip((sc.shs[:, :3] * 0.28209479177387814 + 0.5) * 255, 0, 255).astype(np.uint8)
    root = str(tmp_path / "scene")
    write_scene(root, cams, imgs, sc.pws, rgb, model)
    ds = GSplatDataset(root)
    assert len(ds) == 4 and os.path.exists(os.path.join(root, "sparse", "0", "points3D.npy"))
    cam0, img0 = ds[0]
    assert (cam0.width, cam0.height) == (96, 64) and img0.shape == (3, 64, 96) and img0.dtype == torch.float32
    assert abs(cam0.fx - cams[0].fx) < 1e-9 and abs(cam0.cy - cams[0].cy) < 1e-9
    np.testing.assert_allclose(cam0.Rcw.cpu().numpy(), cams[0].Rcw, atol=1e-6)
    np.testing.assert_allclose(img0.permute(1, 2, 0).cpu().numpy(), imgs[0] / 255.0, atol=1e-7)
    twc = np.stack([-np.linalg.inv(c.Rcw) @ c.tcw for c in cams])
    want_size = 1.1 * np.linalg.norm(twc - twc.mean(0), axis=1).max()
    assert abs(ds.sence_size - want_size) < 1e-4 * want_size
    assert ds.gs.shape == (2000,) and np.allclose(ds.gs["alpha"], 0.8)
    half = GSplatDataset(root, resize_rate=0.5)
    c_half, i_half = half[1]
    assert i_half.shape == (3, 32, 48) and abs(c_half.fx - 0.5 * cams[1].fx) < 1e-9
    # train.py counterpart: initial Gaussians from the point cloud, a few optimizer steps
    start = S.Scene(ds.gs["pw"].copy(), ds.gs["rot"].copy(), ds.gs["scale"].copy(), ds.gs["alpha"].copy(),
                    ds.gs["sh"].copy(), cams[0])
    tr = Trainer(start, ds.cameras, ds.images, max_steps=100, scene_size=ds.sence_size)
    losses = [tr.step([0, 1, 2, 3]) for _ in range(25)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    if model == "PINHOLE":      # the train.py counterpart script end to end
        import subprocess
        import sys
        from tests.conftest import REPO
        out = str(tmp_path / "ckpt")
        r = subprocess.run([sys.executable, os.path.join(REPO, "examples", "train.py"), "--path", root, "--epochs", "4",
                            "--out", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Training is finished." in r.stdout and "epoch:3 avg_loss:" in r.stdout
        final = np.load(os.path.join(out, "final.npy"))
        assert final.dtype == np.dtype(S.gsdata_type(48)) and np.isfinite(final["pw"]).all()


def _gl_matrices(width, height, fov_deg=60.0, near=0.1, far=100.0):
    """A view matrix looking down -z from (0.5, -0.3, 6) and an OpenGL perspective matrix, mathematical
    (row-major) convention as gaussian_item.py holds them."""
    t = np.tan(np.radians(fov_deg) / 2)
    P = np.array([[1 / (t * width / height), 0, 0, 0], [0, 1 / t, 0, 0],
                  [0, 0, -(far + near) / (far - near), -2 * far * near / (far - near)], [0, 0, -1, 0]], np.float64)
    a = 0.3
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    V = np.eye(4)
    V[:3, :3] = R
    V[:3, 3] = -R @ np.array([0.5, -0.3, 6.0])
    focal = (P[0, 0] * width / 2, P[1, 1] * height / 2)            # gaussian_item.py:143-144
    return V, P, focal


@pytest.mark.parametrize("K", [48, 12, 3])
def test_viewer_prep_matches_shader_restatement(K):
    """egs_viewer_prep against the float64 restatement of viewer/shaders/gau_prep.glsl."""
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.viewer import gau_prep, pack_gs_data
    sc = S.small_scene(4000, 320, 200, K, seed=8)
    gs = np.rec.fromarrays([sc.pws, sc.rots, sc.scales, sc.alphas, sc.shs], dtype=S.gsdata_type(K))
    data = pack_gs_data(gs)
    data[:40, 2] += 30.0                                           # behind the GL camera / out of the frustum
    data[40:80, 0] += 20.0
    V, P, focal = _gl_matrices(320, 200)
    prep, depth = gau_prep(data, V, P, focal)
    want, wdepth, culled = io_oracle.viewer_prep(data, V, P, focal)
    assert 50 < culled.sum() < 3000
    prep, depth = prep.cpu().numpy(), depth.cpu().numpy()
    np.testing.assert_allclose(depth, wdepth, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(prep[:, 0] == -100, culled)
    np.testing.assert_array_equal(prep[culled], want[culled])
    ok = ~culled
    scale = np.maximum(1.0, np.abs(want[ok]))
    assert (np.abs(prep[ok] - want[ok]) / scale).max() < 2e-4
    # a record array is accepted as well
    prep2, _ = gau_prep(gs, V, P, focal)
    assert prep2.shape == (4000, 12)


def test_forward_gpu_script_counterpart(tmp_path):
    """examples/forward_gpu.py (the reference's forward_gpu.py on the drop-in module): example scene and a
    .ply written by gau_io."""
    import subprocess
    import sys
    from PIL import Image
    from easygaussiansplatting_amd import gau_io
    from tests.conftest import REPO
    script = os.path.join(REPO, "examples", "forward_gpu.py")
    out = str(tmp_path / "a.png")
    r = subprocess.run([sys.executable, script, "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    img = np.asarray(Image.open(out))
    assert img.shape == (546, 979, 3) and img.max() > 100            # the four blobs are there
    ply = str(tmp_path / "ex.ply")
    gau_io.save_ply(ply, gau_io.get_example_gs())
    out2 = str(tmp_path / "b.png")
    r = subprocess.run([sys.executable, script, "--gs", ply, "--out", out2, "--policy", "forward_cpu"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    img2 = np.asarray(Image.open(out2)).astype(int)
    assert np.abs(img2 - img.astype(int)).mean() < 2.0                # same picture under the CPU-path semantics


def test_reference_module_names_carry_a_train_py_style_loop(tmp_path):
    """The loop of the reference's train.py:30-83 written against the reference's OWN module names
    (gsplat.gsmodel / gsplat.pytorch_ssim / gsplat.gau_io / gsplat.gausplat_dataset, torch.optim.Adam):
    with ``<repo>/compat`` on the path (the opt-in of INTEGRATION.md 1b) those names resolve to the MI355X
    implementations."""
    import sys
    import torch.optim as optim
    from tests.conftest import REPO
    compat = os.path.join(REPO, "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    for k in [k for k in sys.modules if k == "gsplat" or k.startswith("gsplat.")]:
        del sys.modules[k]
    from easygaussiansplatting_amd import gsplatcu as gsc
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera, render
    from tests.colmap_fixture import write_scene
    from gsplat.pytorch_ssim import gau_loss, ssim
    from gsplat.gau_io import save_training_params, load_gs
    from gsplat.gausplat_dataset import GSplatDataset
    from gsplat.gsmodel import GSModel, get_training_params
    gsc.set_policy("gsplatcu")
    sc = S.small_scene(2000, 96, 64, 3, seed=4)
    cams = S.ring_cameras(sc.cam, 4, radius=5.0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    imgs = []
    with torch.no_grad():
        for c in cams:
            im = render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), Camera.from_scene(c))[0]
            imgs.append((im.clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255 + 0.5).astype(np.uint8))
    rgb = np.clip((sc.shs[:, :3] * 0.28209479177387814 + 0.5) * 255, 0, 255).astype(np.uint8)
    root = str(tmp_path / "scene")
    write_scene(root, cams, imgs, sc.pws, rgb)

    gs_set = GSplatDataset(root)
    training_params, adam_params = get_training_params(gs_set.gs)
    optimizer = optim.Adam(adam_params, lr=0.000, eps=1e-15)
    epochs, n = 8, len(gs_set)
    model = GSModel(gs_set.sence_size, len(gs_set) * epochs)
    model.grad_threshold = 1e-7                       # tiny images: let the densification fire
    history = []
    for epoch in range(epochs):
        idxs = np.arange(n)
        np.random.default_rng(epoch).shuffle(idxs)
        avg_loss = 0
        for i in idxs:
            cam, image_gt = gs_set[i]
            image = model(*training_params.values(), cam)
            loss = gau_loss(image, image_gt)
            loss.backward()
            model.update_density_info()
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
            model.update_pws_lr(optimizer)
            avg_loss += loss.item()
        history.append(avg_loss / n)
        with torch.no_grad():
            if epoch == 3:
                model.update_gaussian_density(training_params, optimizer)
            if epoch == 5:
                model.reset_alpha(training_params, optimizer)
    assert np.isfinite(history).all() and history[2] < history[0]
    assert training_params["pws"].shape[0] != 2000 and model.iteration == epochs * n
    fn = str(tmp_path / "final.npy")
    save_training_params(fn, training_params)
    back = load_gs(fn)
    assert back.shape[0] == training_params["pws"].shape[0] and back["sh"].shape[1] == 48
    s = float(ssim(image.detach(), image_gt))
    assert 0.0 < s <= 1.0
