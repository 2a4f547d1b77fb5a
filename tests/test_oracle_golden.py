"""Pin the CPU oracle (oracle/gs_oracle.py) against golden vectors produced by
the reference's own Python (tests/golden/make_golden.py, fixtures G1-G6 of
SURVEY.md §8c).  CPU-only."""
import numpy as np
import pytest

from oracle import gs_oracle as O
from easygaussiansplatting_amd import scene as S
from tests.conftest import load_golden, ref_check

TIGHT = dict(rtol=1e-9, atol=1e-10)


def _cam_from(g):
    fx, fy, cx, cy, W, H = g["intr"]
    return S.Camera(int(W), int(H), fx, fy, cx, cy, g["Rcw"], g["tcw"])


# ------------------------------------------------------------------ G1: oracle B stages
def test_g1_stages_policy_b():
    g = load_golden("g1_stages_b.npz")
    cam = _cam_from(g)
    P = O.POLICY_B
    us, pcs, depths, du = O.project(g["pws"], cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, P, True)
    np.testing.assert_allclose(pcs, g["pcs"], **TIGHT)
    np.testing.assert_allclose(us, g["us"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(du, g["du_dpcs"], rtol=1e-9, atol=1e-7)
    c3, dq, ds = O.compute_cov3d(g["rots"], g["scales"], depths, P, True)
    np.testing.assert_allclose(c3, g["cov3ds"], **TIGHT)
    np.testing.assert_allclose(dq, g["dcov3d_drots"], **TIGHT)
    np.testing.assert_allclose(ds, g["dcov3d_dscales"], **TIGHT)
    c2, d3, dpc = O.compute_cov2d(g["cov3ds"], g["pcs"], cam.Rcw, depths, cam.fx, cam.fy, cam.width,
                                  cam.height, P, True)
    np.testing.assert_allclose(c2, g["cov2ds"], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(d3, g["dcov2d_dcov3ds"], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(dpc, g["dcov2d_dpcs"], rtol=1e-8, atol=1e-5)
    ci, areas, dci = O.inverse_cov2d(g["cov2ds"], depths, P, True)
    np.testing.assert_allclose(ci, g["cinv2ds"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dci, g["dcinv2d_dcov2ds"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("K", [3, 12, 27, 48])
def test_g1_sh2color_all_degrees(K):
    g = load_golden("g1_stages_b.npz")
    col, dsh, dpw = O.sh2color(g["shs"][:, :K], g["pws"], g["twc"], True)
    sfx = "" if K == 48 else "_K%d" % K
    np.testing.assert_allclose(col, g["colors" + sfx], **TIGHT)
    np.testing.assert_allclose(dsh, g["dcolor_dshs" + sfx], **TIGHT)
    np.testing.assert_allclose(dpw, g["dcolor_dpws" + sfx], rtol=1e-9, atol=1e-10)


def test_g1_known_answers():
    """Inputs of the reference's test/test_cov3d.py:112-113 and test_cov2d.py:104-110
    (values also listed in SURVEY.md appendix)."""
    g = load_golden("g1_stages_b.npz")
    c3, dq, ds = O.compute_cov3d(g["ka_q"][None], g["ka_s"][None], None, O.POLICY_B, True)
    np.testing.assert_allclose(c3[0], [1.24892526, -2.73532296, 0.86639549, 7.97665233, -3.00404921, 2.70966732],
                               atol=1e-7)
    np.testing.assert_allclose(c3[0], g["ka_cov3d"], **TIGHT)
    np.testing.assert_allclose(dq[0], g["ka_dcov3d_dq"], **TIGHT)
    np.testing.assert_allclose(ds[0], g["ka_dcov3d_ds"], **TIGHT)
    c2, d3, dpc = O.compute_cov2d(g["ka_cov3d"][None], g["ka_pc"][None], g["ka_Rcw"], None, 200.0, 100.0,
                                  0, 0, O.POLICY_B, True)
    np.testing.assert_allclose(c2[0] - [0.3, 0, 0.3], [9341.30334186, 9107.59766184, 16025.72770731], atol=1e-6)
    np.testing.assert_allclose(c2[0], g["ka_cov2d"], rtol=1e-12)
    np.testing.assert_allclose(d3[0], g["ka_dcov2d_dcov3d"], rtol=1e-12)
    np.testing.assert_allclose(dpc[0], g["ka_dcov2d_dpc"], rtol=1e-10)


# ------------------------------------------------------------------ G2: oracle A stages
def test_g2_stages_policy_a():
    g1 = load_golden("g1_stages_b.npz")
    g = load_golden("g2_stages_a.npz")
    cam = _cam_from(g1)
    P = O.POLICY_A
    us, pcs, depths = O.project(g1["pws"], cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, P)
    np.testing.assert_allclose(pcs, g["pcs"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(us, g["us"], rtol=1e-10, atol=1e-8)
    # the reference builds R from float32 quaternions in float32 (gausplat.py:116-120):
    # float32-level agreement only
    c3 = O.compute_cov3d(g1["rots"], g1["scales"], depths, P)
    np.testing.assert_allclose(c3, g["cov3ds"], rtol=2e-6, atol=1e-8)
    with np.errstate(all="ignore"):
        c2 = O.compute_cov2d(g["cov3ds"], g["pcs"], cam.Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, P)
    fin = np.isfinite(g["cov2ds"]).all(1)
    np.testing.assert_allclose(c2[fin], g["cov2ds"][fin], rtol=1e-9, atol=1e-6)
    # A's fov limit is 1.3 * 2*atan(W/2fx), not the tangent
    lim = O.fov_limits(cam.fx, cam.fy, cam.width, cam.height, P)
    assert abs(lim[0] - 1.3 * 2 * np.arctan(cam.width / (2 * cam.fx))) < 1e-12
    # float32 SH products in the reference (gausplat.py:58): float32-level agreement
    col = O.sh2color(g1["shs"], g1["pws"], g1["twc"])
    np.testing.assert_allclose(col, g["colors"], rtol=0, atol=2e-6)
    ci, areas = O.inverse_cov2d(g["cov2ds"], depths, P)
    np.testing.assert_allclose(ci[fin], g["cinv2ds"][fin], rtol=1e-9, atol=1e-12)
    # int32 radii = trunc(3 sqrt(a)) (gausplat.py:181-182): bit-exact where defined
    okr = fin & (g["cov2ds"][:, 0] > 0) & (g["cov2ds"][:, 2] > 0) & (np.abs(g["areas"]) < 2**30).all(1)
    assert np.array_equal(areas[okr], g["areas"][okr])


# ------------------------------------------------------------------ G3: backward_gpu.py's comparison set
def _g3_pipeline(g, dtype=np.float64):
    sc = S.example_gs()
    cam = sc.cam
    P = O.POLICY_G
    fake_depths = np.array([1, 2, 3, 4], dtype)          # backward_gpu.py:87-88
    us, pcs, _, du = O.project(g["pws"], cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, P, True, dtype)
    c3, dq, ds = O.compute_cov3d(g["rots"], g["scales"], fake_depths, P, True, dtype)
    c2, d3, dpc = O.compute_cov2d(c3, pcs, cam.Rcw, fake_depths, cam.fx, cam.fy, cam.width, cam.height,
                                  P, True, dtype)
    col, dsh, dpw = O.sh2color(g["shs"], g["pws"], cam.twc, True, dtype)
    ci, areas, dci = O.inverse_cov2d(c2, fake_depths, P, True, dtype)
    return dict(us=us, pcs=pcs, du_dpcs=du, cov3ds=c3, dcov3d_drots=dq, dcov3d_dscales=ds, cov2ds=c2,
                dcov2d_dcov3ds=d3, dcov2d_dpcs=dpc, colors=col, dcolor_dshs=dsh, dcolor_dpws=dpw,
                cinv2ds=ci, dcinv2d_dcov2ds=dci, areas=areas, depths=fake_depths, cam=cam)


def test_g3_example_scene_everything_backward_gpu_checks():
    g = load_golden("g3_example_backward.npz")
    r = _g3_pipeline(g)
    cam = r["cam"]
    for k in ("us", "pcs", "du_dpcs", "cov3ds", "dcov3d_drots", "dcov3d_dscales", "cov2ds", "dcov2d_dcov3ds",
              "dcov2d_dpcs", "colors", "dcolor_dshs", "dcolor_dpws", "cinv2ds", "dcinv2d_dcov2ds"):
        assert ref_check(r[k], g[k], 1e-9), k
    # appendix facts: gsplatcu radii and P = 5 patches over 2x1 tiles
    assert r["areas"].tolist() == [[2, 2], [3, 2], [2, 3], [2, 2]]
    image, contrib, tau, ranges, gsid = O.splat(cam.height, cam.width, r["us"], r["cinv2ds"], g["alphas"],
                                                r["depths"], r["colors"], r["areas"], O.POLICY_G)
    assert gsid.shape[0] == 5 and ranges.shape == (2, 2)
    assert ref_check(image.transpose(1, 2, 0), g["image"], 1e-12)
    # L1 loss vs zeros: dL/dimage = sign(image)/numel
    dl = np.sign(image) / image.size
    np.testing.assert_allclose(dl, g["dloss_dgammas"], atol=1e-15)
    dus, dci, da, dc = O.draw_backward(cam.width, cam.height, ranges, gsid, r["us"], r["cinv2ds"], g["alphas"],
                                       r["colors"], contrib, tau, g["dloss_dgammas"], None, O.POLICY_G)
    assert ref_check(dus[:, None], g["dloss_dus"], 1e-10)
    assert ref_check(dci[:, None], g["dloss_dcinv2ds"], 1e-10)
    assert ref_check(da[:, None, None], g["dloss_dalphas"], 1e-10)
    assert ref_check(dc[:, None], g["dloss_dcolors"], 1e-10)
    J = {k: r[k] for k in ("dcinv2d_dcov2ds", "dcov2d_dcov3ds", "dcov3d_drots", "dcov3d_dscales",
                           "dcolor_dshs", "du_dpcs", "dcov2d_dpcs", "dcolor_dpws")}
    gr = O.chain_rule(dus, dci, da, dc, cam.Rcw, J)
    assert ref_check(gr["drots"][:, None], g["dloss_drots"], 1e-10)
    assert ref_check(gr["dscales"][:, None], g["dloss_dscales"], 1e-10)
    assert ref_check(gr["dshs"][:, None], g["dloss_dshs"], 1e-10)
    assert ref_check(gr["dpws"][:, None], g["dloss_dpws"], 1e-10)
    assert ref_check(gr["dalphas"][:, None, None], g["dloss_dalphas_final"], 1e-10)


def test_g3_float32_oracle_within_reference_tolerance():
    """The same pipeline evaluated in float32 stays inside the reference's
    1e-4 acceptance band (what the HIP path must achieve)."""
    g = load_golden("g3_example_backward.npz")
    r = _g3_pipeline(g, np.float32)
    for k in ("us", "cov2ds", "cinv2ds", "colors", "dcov2d_dpcs"):
        assert ref_check(r[k], g[k], 1e-4), k


# ------------------------------------------------------------------ G4: forward_cpu.py, BASELINE configs[0]
def test_g4_forward_cpu_10k_policy_a_full_pipeline():
    g = load_golden("g4_forward_cpu_10k.npz")
    sc = S.small_scene()
    out = O.forward_pipeline((sc.pws, sc.rots, sc.scales, sc.alphas, sc.shs), sc.cam, O.POLICY_A)
    k = 64
    np.testing.assert_allclose(out["us"][:k], g["us"], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(out["cov2ds"][:k], g["cov2ds"], rtol=2e-5, atol=1e-6)  # f32 R in the reference
    np.testing.assert_allclose(out["colors"][:k], g["colors"], rtol=0, atol=2e-6)
    assert np.array_equal(out["areas"][:k], g["areas"])
    img = out["image"].transpose(1, 2, 0)
    d = np.abs(img - g["image"])
    # stated tolerance of the path: 1e-4 abs on the image
    assert d.max() < 1e-4, d.max()


def test_g4_loop_restatement_is_bit_identical_given_same_2d_records():
    """splat_forward_cpu (the cpu_baseline) and the tile formulation of policy
    A are the same function of the 2D records."""
    sc = S.small_scene(3000)
    out = O.forward_pipeline((sc.pws, sc.rots, sc.scales, sc.alphas, sc.shs), sc.cam, O.POLICY_A)
    img = O.splat_forward_cpu(sc.cam.height, sc.cam.width, out["us"], out["cinv2ds"],
                              sc.alphas.astype(np.float64), out["depths"], out["colors"], out["areas"])
    assert np.abs(img.transpose(2, 0, 1) - out["image"]).max() < 1e-13


# ------------------------------------------------------------------ G5: oracle B raster, multi-tile
def test_g5_raster_forward_backward_policy_g():
    g = load_golden("g5_raster_b_multitile.npz")
    W, H = 48, 32
    depths = g["depths"].copy(); areas = g["areas"].copy()
    ranges, gsid, rects, counts = O.bin_tiles(g["us"], areas, depths, W, H, O.POLICY_G)
    assert np.array_equal(ranges, g["ranges"]) and np.array_equal(gsid, g["gsid"])
    image, contrib, tau = O.draw(W, H, ranges, gsid, g["us"], g["cinv2ds"], g["alphas"], g["colors"],
                                 None, O.POLICY_G)
    assert np.abs(image.transpose(1, 2, 0) - g["image"]).max() < 1e-12
    assert np.array_equal(contrib, g["contrib"])
    np.testing.assert_allclose(tau, g["final_tau"], rtol=1e-10, atol=1e-15)
    assert (tau[contrib > 0] < 1e-4).any(), "fixture must exercise the tau<1e-4 early stop"
    dus, dci, da, dc = O.draw_backward(W, H, ranges, gsid, g["us"], g["cinv2ds"], g["alphas"], g["colors"],
                                       contrib, tau, g["dloss_dgammas"], None, O.POLICY_G)
    for a, b in ((dus, g["dloss_dus"]), (dci, g["dloss_dcinv2ds"]), (da, g["dloss_dalphas"]),
                 (dc, g["dloss_dcolors"])):
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-12)


# ------------------------------------------------------------------ G6: forward_cpu.py at BASELINE size (digest)
@pytest.mark.slow
def test_g6_digest_sampled_tiles_policy_a():
    """64 full-resolution tiles of the reference's 1M/1080p image, recomputed
    by the oracle (policy A) for those tiles only."""
    import os
    from tests.conftest import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "g6_forward_cpu_1m_digest.npz")):
        pytest.skip("G6 digest not generated")
    g = load_golden("g6_forward_cpu_1m_digest.npz")
    sc = S.big_scene()
    cam = sc.cam
    P = O.POLICY_A
    us, pcs, depths = O.project(sc.pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, P)
    c3 = O.compute_cov3d(sc.rots, sc.scales, depths, P)
    c2 = O.compute_cov2d(c3, pcs, cam.Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, P)
    col = O.sh2color(sc.shs, sc.pws, cam.twc)
    ci, areas = O.inverse_cov2d(c2, depths, P)
    x0, x1, y0, y1 = O.pixel_box(us, areas, cam.width, cam.height)
    gx, gy = O.tile_grid(cam.width, cam.height)
    order = np.argsort(depths, kind="stable")
    for t, ref_tile in zip(g["tile_ids"][:16], g["tiles"][:16]):
        ty, tx = divmod(int(t), gx)
        hit = (x0 < tx * 16 + 16) & (x1 > tx * 16) & (y0 < ty * 16 + 16) & (y1 > ty * 16)
        ids = order[hit[order]]
        ranges = np.zeros((gx * gy, 2), np.int32); ranges[t] = (0, ids.size)
        img, _, _ = O.draw(cam.width, cam.height, ranges, ids.astype(np.int32), us, ci, sc.alphas, col,
                           areas, P, tiles=[int(t)])
        tile = img[:, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].transpose(1, 2, 0)
        assert np.abs(tile - ref_tile).max() < 1e-4


# ------------------------------------------------------------------ G7: training loss (pytorch_ssim.gau_loss)
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g7_gau_loss_and_gradient(tag):
    g = load_golden("g7_gau_loss.npz")
    loss, grad, aux = O.gau_loss(g["x_" + tag], g["y_" + tag], 0.2, calc_grad=True)
    np.testing.assert_allclose(loss, g["loss_" + tag], rtol=1e-7)        # window built in float32 in the reference
    np.testing.assert_allclose(aux["ssim"], g["ssim_" + tag], rtol=1e-7)
    scale = np.abs(g["grad_" + tag]).max()
    assert np.abs(grad - g["grad_" + tag]).max() < 1e-6 * scale
    assert (grad[:, : g["x_" + tag].shape[1] // 3] != 0).any()          # SSIM part acts where |x-y| = 0


def test_g10_nan_conic_semantics():
    """Fixture G10: the CUDA extension's max(0.f, NaN) = 0 blends a Gaussian with a NaN Mahalanobis term at
    min(0.99, alpha) (kernel.cu:243-246, NAN_MAHA = "cuda"); the build's default does the same for a NaN IN the conic
    and skips the inf * 0 pixels of an infinite one ("entry"); the opt-in policy skips every NaN pixel ("skip")."""
    g = load_golden("g10_nan_conic.npz")
    W, H = int(g["width"]), int(g["height"])
    try:
        for mode in ("skip", "entry", "cuda"):
            O.NAN_MAHA = mode
            with np.errstate(all="ignore"):
                img, cont, tau, ranges, gsid = O.splat(H, W, g["us"], g["cinv2ds"], g["alphas"].astype(np.float64),
                                                       g["depths"].copy(), g["colors"], g["areas"].copy(), O.POLICY_G)
            assert np.abs(img - g["image_" + mode]).max() < 1e-6 and np.array_equal(cont, g["contrib_" + mode])
    finally:
        O.NAN_MAHA = "cuda"      # the oracle default: the reference's arithmetic
    # skip == the two finite Gaussians alone
    keep = np.array([0, 3])
    img2 = O.splat(H, W, g["us"][keep], g["cinv2ds"][keep], g["alphas"][keep].astype(np.float64), g["depths"][keep].copy(),
                   g["colors"][keep], g["areas"][keep].copy(), O.POLICY_G)[0]
    assert np.abs(img2 - g["image_skip"]).max() < 1e-6
    # cuda: Gaussian 2 (NaN conic, alpha 0.5, in front of 0 and 3) tints EVERY pixel of both tiles
    assert (np.abs(g["image_cuda"] - g["image_skip"]).max(0) > 1e-3).all()
    # entry == cuda but for the pixel column through the centre of the infinite conic (u.x = 20: inf * 0)
    d = np.abs(g["image_entry"] - g["image_cuda"]).max(0)
    assert set(np.nonzero(d > 1e-6)[1]) == {20}
