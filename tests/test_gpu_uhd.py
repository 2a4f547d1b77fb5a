"""3840 x 2160 once.  The reference is resolution-agnostic (kernel.cu:152, grid from gausplat.cu:94); here
T = 32 400 tiles need 15 tile-key bits (two 8-bit passes of the tile sort instead of 7 + 6) and the tail registers of
``k_tile_order``; nothing above 1080p had run on hardware before round 6.  The bench scene through a camera of twice the
focal length: seven-op lists BIT-EXACT against ``O.bin_tiles``, sampled tiles of the image against ``O.draw``, the four
``splatB`` gradients on a window against ``O.draw_backward`` (default rule), and the fused path against both."""
import numpy as np
import pytest

from easygaussiansplatting_amd import scene as S
from oracle import gs_oracle as O
from tests.gradcheck import assert_grad_close_flips
from tests.test_gpu_parity import complete_inside, dev, host, window_tiles

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_uhd_lists_image_and_gradients_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easygaussiansplatting_amd import fused, gsplatcu as gsc
    from easygaussiansplatting_amd.function import Camera, GSFunction
    from tests.test_gpu_round5_vs_oracle import stages
    gsc.set_policy("gsplatcu")
    W, H = 3840, 2160
    sc = S.big_scene(600_000, W, H, 12)
    sc.cam = S.Camera(W, H, 2400.0, 2400.0, W / 2.0, H / 2.0, sc.cam.Rcw, sc.cam.tcw)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    assert gx * gy == 32400
    g = stages(gsc, sc)
    d, a = g["depths"].clone(), g["areas"].clone()
    image, contrib, tau, ranges, gsid = gsc.splat(H, W, g["us"], g["cinv"], g["alphas"], d, g["col"], a)
    o_d, o_a = host(g["depths"]).copy(), host(g["areas"]).copy()
    o_rg, o_gs, _, _ = O.bin_tiles(host(g["us"]), o_a, o_d, W, H, O.POLICY_G)
    rg, gs = host(ranges), host(gsid)
    assert np.array_equal(rg, o_rg) and np.array_equal(gs, o_gs)          # createKeys / sort / getRanges, 15 tile bits
    assert gs.shape[0] > 4_000_000
    hu, hc, ha, hcol = host(g["us"]), host(g["cinv"]), host(g["alphas"]), host(g["col"])
    lens = rg[:, 1] - rg[:, 0]
    tl = int(np.argmax(lens))
    sel = np.unique(np.concatenate([[0, gx - 1, (gy - 1) * gx, gx * gy - 1, tl],
                                    (S.uniform01(4, 5, (11,)) * gx * gy).astype(np.int64)]))
    o_img, o_cont, o_tau = O.draw(W, H, rg, gs, hu, hc, ha, hcol, None, O.POLICY_G, tiles=sel)
    him, hcont, htau = host(image), host(contrib), host(tau)
    nflip = 0
    for t in sel:
        ty, tx = divmod(int(t), gx)
        ys = slice(ty * 16, ty * 16 + 16); xs = slice(tx * 16, tx * 16 + 16)
        e = np.abs(him[:, ys, xs] - o_img[:, ys, xs]).max(0)
        flip = (hcont[ys, xs] != o_cont[ys, xs]) | (e >= 1e-4)
        nflip += int(flip.sum())
        assert e[~flip].max() < 1e-4 and e.max() < 5e-3 and np.abs(htau[ys, xs] - o_tau[ys, xs])[~flip].max() < 1e-4
    assert nflip <= 8, nflip
    dl = S.normal(8, 3, (3, H, W)).astype(np.float32) / (H * W)
    grads = gsc.splatB(H, W, g["us"], g["cinv"], g["alphas"], d, g["col"], contrib, tau, ranges, gsid, dev(dl))
    sub = np.unique(np.concatenate([window_tiles(gx, gy, tl % gx, tl // gx), window_tiles(gx, gy, gx // 2, gy - 1)]))
    near = np.zeros(sc.n, bool)
    o_g = O.draw_backward(W, H, rg, gs, hu, hc, ha, hcol, hcont, htau, dl, None, O.POLICY_G, tiles=sub, near_out=near)
    full = complete_inside(gs, rg, sub, sc.n)
    assert full.size > 300, full.size
    for x, y, nm in zip(o_g, grads, ("dus", "dcinv", "dalpha", "dcolor")):
        y = host(y).reshape(x.shape)
        assert_grad_close_flips(y[full], x[full], near[full], "uhd_ops:" + nm)
    # the fused training op at the same size: same image up to the rounding of its own float32 stages, and its dL/dsh
    # (degree 0) = the oracle-checked dL/dcolour of the seven-op path x Y_00 (sh2Color, kernel.cu:619-807)
    GSFunction.mode = "fused"
    P = [dev(sc.pws), dev(sc.shs), dev(sc.alphas).reshape(-1, 1), dev(sc.scales), dev(sc.rots)]
    for p in P:
        p.requires_grad_(True)
    us0 = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    with fused.deferred() as df:
        img_f, _ = GSFunction.apply(*P, us0, Camera.from_scene(sc.cam))
        img_f.backward(dev(dl))
        assert not df.commit()
    e = np.abs(host(img_f) - him).max(0)
    assert (e >= 2e-5).mean() < 2e-5 and e.max() < 5e-3, ((e >= 2e-5).sum(), e.max())
    sh0 = host(P[1].grad)[:, :3]
    want = host(grads[3]).reshape(-1, 3) * 0.28209479177387814
    assert np.abs(sh0 - want).max() < 2e-4 * np.abs(want).max()
    assert all(torch.isfinite(p.grad).all() for p in P)


def test_uhd_segment_path_equals_unsplit_kernels():
    """The long-list split at 3840 x 2160 (32 400 tiles: beyond the dispatch-order kernel's table, 15 tile bits, 127 plan
    workgroups): ``scene.skewed_scene`` right after ``reset_alpha`` through a camera of twice the focal length (lists to
    ~8 300 entries, walks to ~4 400) -- segment path == unsplit kernels: lists, image, contributor counts, final
    transmittance and all parameter gradients (tests/test_gpu_segments.py's comparison, default tolerance)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easygaussiansplatting_amd import fused, gsplatcu as gsc
    from easygaussiansplatting_amd.function import Camera
    from tests.test_gpu_segments import compare, run
    gsc.set_policy("gsplatcu")
    W, H = 3840, 2160
    sc = S.skewed_scene(reset_alpha=True)
    sc.cam = S.Camera(W, H, 2 * sc.cam.fx, 2 * sc.cam.fy, W / 2.0, H / 2.0, sc.cam.Rcw, sc.cam.tcw)
    dl = dev(S.normal(3, 22, (3, H, W)).astype(np.float32) / (3 * H * W))
    keep = fused.SEGMENTS
    try:
        fused.SEGMENTS = "0"
        ref = run(fused, sc, Camera.from_scene(sc.cam), dl)
        fused.SEGMENTS = "auto"
        got = run(fused, sc, Camera.from_scene(sc.cam), dl, 2)
    finally:
        fused.SEGMENTS = keep
    lens = ref["ranges"][:, 1] - ref["ranges"][:, 0]
    assert lens.size == 32400 and lens.max() > 5000 and ref["contrib"].max() > 3000 and got["seg"] and not ref["seg"]
    compare(got, ref, "uhd_skewed_reset", flips=256)
